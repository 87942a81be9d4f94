// CommonJS build of proof.helper.ts (Node >= 12 without a TypeScript runner) -- same exports as the reference's
// client/proof.helper.ts: generateProof(config, inputs) -> { proof: Buffer, publicWitness: Buffer }.
"use strict";
const fs = require("fs");
const path = require("path");
const addon = require("./spp_addon.node");

const FIELD_ORDER = ["root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index"];
const handles = new Map();

function toField32(v) {
  let n = typeof v === "bigint" ? v : BigInt(v);          // "0x.." strings, decimal strings and numbers
  const out = Buffer.alloc(32);
  for (let i = 31; i >= 0; i--) { out[i] = Number(n & 0xffn); n >>= 8n; }
  if (n !== 0n) throw new Error("libspp error -1: value does not fit 32 bytes");
  return out;
}

// text of client/proof.helper.ts:32-50 of the reference
function proverToml(inputs) {
  const quoted = new Set(["root", "nullifier", "recipient", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness"]);
  let toml = "";
  for (const k of FIELD_ORDER) toml += quoted.has(k) ? `${k} = "${inputs[k]}"\n` : `${k} = ${inputs[k]}\n`;
  toml += "siblings = [\n";
  for (const sib of inputs.siblings) toml += `  "${sib}",\n`;
  return toml + "]\n";
}

function circuitHandle(config) {
  const key = path.resolve(config.circuitDir) + "/" + config.circuitName;
  if (!handles.has(key)) {
    const target = path.join(config.circuitDir, "target");
    addon.init(parseInt(process.env.SPP_DEVICE || "0", 10));
    handles.set(key, addon.loadCircuit(path.join(target, `${config.circuitName}.sppc`), path.join(target, `${config.circuitName}.pk`), 0));
  }
  return handles.get(key);
}

function generateProof(config, inputs) {
  if (!Array.isArray(inputs.siblings) || inputs.siblings.length !== 16) throw new Error("siblings must hold 16 elements");
  fs.writeFileSync(path.join(config.circuitDir, "Prover.toml"), proverToml(inputs));
  const h = circuitHandle(config);
  const parts = FIELD_ORDER.map((k) => toField32(inputs[k])).concat(inputs.siblings.map(toField32));
  const r = addon.proveBatch(h, 1, Buffer.concat(parts), null);
  if (r.status[0] !== 0) throw new Error(`libspp error ${r.status[0]}: inputs do not satisfy the circuit`);
  const target = path.join(config.circuitDir, "target");
  fs.writeFileSync(path.join(target, `${config.circuitName}.proof`), r.proofs);
  fs.writeFileSync(path.join(target, `${config.circuitName}.pw`), r.publicWitnesses);
  return { proof: r.proofs, publicWitness: r.publicWitnesses };
}

module.exports = { generateProof, proverToml, toField32, addon };
