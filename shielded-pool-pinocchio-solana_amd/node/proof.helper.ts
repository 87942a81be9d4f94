// Drop-in for the reference's client/proof.helper.ts: same exported types and function, but the proof comes from
// libspp (HIP, MI355X) through the N-API addon instead of `nargo execute` + `sunspot prove` child processes.
// Swap the import path in client/test-shielded-pool.ts:246 / client/payroll-demo.ts:330 and nothing else.
import fs from "fs";
import path from "path";
import { createRequire } from "module";

const require = createRequire(import.meta.url);
const addon = require("./spp_addon.node");

export interface ShieldedPoolInputs {
  // public inputs
  root: string;
  nullifier: string;
  recipient: string;
  amount: number | string;
  wa_commitment: string;
  // private inputs
  secret_key: string;
  owner_x: string;
  owner_y: string;
  randomness: string;
  index: number | string;
  siblings: string[];
}

export interface CircuitConfig {
  circuitDir: string;
  circuitName: string;
}

const FIELD_ORDER = ["root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index"] as const;
const QUOTED = new Set<string>(["root", "nullifier", "recipient", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness"]);
const handles = new Map<string, unknown>();

function toField32(v: string | number | bigint): Buffer {
  let n = typeof v === "bigint" ? v : BigInt(v);
  const out = Buffer.alloc(32);
  for (let i = 31; i >= 0; i--) { out[i] = Number(n & 0xffn); n >>= 8n; }
  if (n !== 0n) throw new Error("libspp error -1: value does not fit 32 bytes");
  return out;
}

export function proverToml(inputs: ShieldedPoolInputs): string {
  let toml = "";
  for (const k of FIELD_ORDER) toml += QUOTED.has(k) ? `${k} = "${inputs[k]}"\n` : `${k} = ${inputs[k]}\n`;
  toml += "siblings = [\n";
  for (const sib of inputs.siblings) toml += `  "${sib}",\n`;
  return toml + "]\n";
}

export function generateProof(config: CircuitConfig, inputs: ShieldedPoolInputs) {
  if (inputs.siblings.length !== 16) throw new Error("siblings must hold 16 elements");
  fs.writeFileSync(path.join(config.circuitDir, "Prover.toml"), proverToml(inputs));
  const target = path.join(config.circuitDir, "target");
  const key = path.resolve(config.circuitDir) + "/" + config.circuitName;
  if (!handles.has(key)) {
    addon.init(parseInt(process.env.SPP_DEVICE || "0", 10));
    handles.set(key, addon.loadCircuit(path.join(target, `${config.circuitName}.sppc`), path.join(target, `${config.circuitName}.pk`), 0));
  }
  const parts = FIELD_ORDER.map((k) => toField32(inputs[k])).concat(inputs.siblings.map(toField32));
  const r = addon.proveBatch(handles.get(key), 1, Buffer.concat(parts), null);
  if (r.status[0] !== 0) throw new Error(`libspp error ${r.status[0]}: inputs do not satisfy the circuit`);
  fs.writeFileSync(path.join(target, `${config.circuitName}.proof`), r.proofs);
  fs.writeFileSync(path.join(target, `${config.circuitName}.pw`), r.publicWitnesses);
  return { proof: r.proofs as Buffer, publicWitness: r.publicWitnesses as Buffer };
}
