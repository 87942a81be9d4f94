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

// Window bits of the MSM tables: the drop-in call proves one statement at a time and does not need the 225 GB of
// wide-window tables of a batch server (0 = auto); 8-bit windows are ~6 GB and build in under a second.  SPP_WINDOW overrides.
function helperWindow(): number {
  const w = parseInt(process.env.SPP_WINDOW || "8", 10);
  return Number.isFinite(w) ? w : 8;
}

function circuitHandle(config: CircuitConfig): unknown {
  const key = path.resolve(config.circuitDir) + "/" + config.circuitName;
  if (!handles.has(key)) {
    const target = path.join(config.circuitDir, "target");
    addon.init(parseInt(process.env.SPP_DEVICE || "0", 10));
    handles.set(key, addon.loadCircuit(path.join(target, `${config.circuitName}.sppc`), path.join(target, `${config.circuitName}.pk`), helperWindow()));
  }
  return handles.get(key);
}

function withdrawRow(inputs: ShieldedPoolInputs): Buffer {
  if (!Array.isArray(inputs.siblings) || inputs.siblings.length !== 16) throw new Error("siblings must hold 16 elements");
  return Buffer.concat(FIELD_ORDER.map((k) => toField32(inputs[k])).concat(inputs.siblings.map(toField32)));
}

function writeOutputs(config: CircuitConfig, proof: Buffer, pw: Buffer): void {
  const target = path.join(config.circuitDir, "target");
  fs.writeFileSync(path.join(target, `${config.circuitName}.proof`), proof);
  fs.writeFileSync(path.join(target, `${config.circuitName}.pw`), pw);
}

export function generateProof(config: CircuitConfig, inputs: ShieldedPoolInputs) {
  const row = withdrawRow(inputs);
  fs.writeFileSync(path.join(config.circuitDir, "Prover.toml"), proverToml(inputs));
  const r = addon.proveBatch(circuitHandle(config), 1, row, null);
  if (r.status[0] !== 0) throw new Error(`libspp error ${r.status[0]}: inputs do not satisfy the circuit`);
  writeOutputs(config, r.proofs, r.publicWitnesses);
  return { proof: r.proofs as Buffer, publicWitness: r.publicWitnesses as Buffer };
}

// Many withdraw proofs in ONE call: the shape client/payroll-demo.ts:326-352 wants from its Promise.all over generateProof
// (which in the reference serialises on execSync and races on the shared Prover.toml).  One { proof, publicWitness } per
// input, in order; throws naming the first index whose inputs do not satisfy the circuit.  No files are written.
export function generateProofBatch(config: CircuitConfig, inputsList: ShieldedPoolInputs[]): { proof: Buffer; publicWitness: Buffer }[] {
  if (inputsList.length === 0) return [];
  const r = addon.proveBatch(circuitHandle(config), inputsList.length, Buffer.concat(inputsList.map(withdrawRow)), null);
  const bad = (r.status as number[]).findIndex((s) => s !== 0);
  if (bad >= 0) throw new Error(`libspp error ${r.status[bad]}: inputs of proof ${bad} do not satisfy the circuit`);
  const pwLen = r.publicWitnesses.length / inputsList.length;
  return inputsList.map((_, i) => ({
    proof: (r.proofs as Buffer).slice(388 * i, 388 * (i + 1)),
    publicWitness: (r.publicWitnesses as Buffer).slice(pwLen * i, pwLen * (i + 1)),
  }));
}

// ---- audit circuit: the reference proves it from scripts (audit_circuit/prove_audit.sh:74-99, scripts/generate_audit.py:668-685)
type FieldLike = string | number | bigint;
export interface AuditInputs {
  // the audit Prover.toml keys, scripts/generate_audit.py:630-641 (signed values allowed: stored as p - |v|, :77-82)
  secret_key: FieldLike;
  wa_commitment: FieldLike;
  ct_commitment: FieldLike;
  c0_packed: FieldLike[]; // 10
  c1_packed: FieldLike[]; // 147
  r: FieldLike[]; // 1024
  e1_sparse: FieldLike[]; // 64
  e2: FieldLike[]; // 1024
  k0: FieldLike[]; // 64
  k1: FieldLike[]; // 1024
}
const BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const AUDIT_ORDER = ["secret_key", "wa_commitment", "ct_commitment", "c0_packed", "c1_packed", "r", "e1_sparse", "e2", "k0", "k1"] as const;
const AUDIT_LEN: Record<string, number> = { c0_packed: 10, c1_packed: 147, r: 1024, e1_sparse: 64, e2: 1024, k0: 64, k1: 1024 };

function toFieldSigned(v: FieldLike): bigint {
  let n = typeof v === "bigint" ? v : BigInt(v);
  n %= BN254_R;
  return n < 0n ? n + BN254_R : n;
}
function formatField(v: FieldLike): string {
  const n = toFieldSigned(v);
  return n === 0n ? '"0"' : `"0x${n.toString(16).padStart(64, "0")}"`;
}
export function auditProverToml(inputs: AuditInputs): string {
  let out = "";
  for (const k of AUDIT_ORDER) {
    const v = inputs[k];
    out += Array.isArray(v) ? `${k} = [${v.map(formatField).join(", ")}]\n` : `${k} = ${formatField(v as FieldLike)}\n`;
  }
  return out;
}
export function generateAuditProof(config: CircuitConfig, inputs: AuditInputs) {
  for (const [k, n] of Object.entries(AUDIT_LEN)) {
    const v = (inputs as unknown as Record<string, unknown>)[k];
    if (!Array.isArray(v) || v.length !== n) throw new Error(`${k} must hold ${n} elements`);
  }
  const order = ["wa_commitment", "ct_commitment", "c0_packed", "c1_packed", "secret_key", "r", "e1_sparse", "e2", "k0", "k1"] as const; // main() :405-417
  const parts: Buffer[] = [];
  for (const k of order) {
    const v = inputs[k];
    for (const x of Array.isArray(v) ? v : [v as FieldLike]) parts.push(toField32(toFieldSigned(x)));
  }
  fs.writeFileSync(path.join(config.circuitDir, "Prover.toml"), auditProverToml(inputs));
  const r = addon.proveBatch(circuitHandle(config), 1, Buffer.concat(parts), null);
  if (r.status[0] !== 0) throw new Error(`libspp error ${r.status[0]}: inputs do not satisfy the circuit`);
  writeOutputs(config, r.proofs, r.publicWitnesses);
  return { proof: r.proofs as Buffer, publicWitness: r.publicWitnesses as Buffer };
}
