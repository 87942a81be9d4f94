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

// Window bits of the MSM tables.  The drop-in call proves ONE statement at a time, so it does not need the 225 GB of
// wide-window tables a batch server builds (window 0 = auto): 8-bit windows are ~6 GB / under a second to build and cost a
// single proof nothing measurable (its latency is the witness solver).  SPP_WINDOW overrides (0 = auto).
function helperWindow() {
  const w = parseInt(process.env.SPP_WINDOW || "8", 10);
  return Number.isFinite(w) ? w : 8;
}

function circuitHandle(config) {
  const key = path.resolve(config.circuitDir) + "/" + config.circuitName;
  if (!handles.has(key)) {
    const target = path.join(config.circuitDir, "target");
    addon.init(parseInt(process.env.SPP_DEVICE || "0", 10));
    handles.set(key, addon.loadCircuit(path.join(target, `${config.circuitName}.sppc`), path.join(target, `${config.circuitName}.pk`), helperWindow()));
  }
  return handles.get(key);
}

function withdrawRow(inputs) {
  if (!Array.isArray(inputs.siblings) || inputs.siblings.length !== 16) throw new Error("siblings must hold 16 elements");
  return Buffer.concat(FIELD_ORDER.map((k) => toField32(inputs[k])).concat(inputs.siblings.map(toField32)));
}

function writeOutputs(config, proof, pw) {
  const target = path.join(config.circuitDir, "target");
  fs.writeFileSync(path.join(target, `${config.circuitName}.proof`), proof);
  fs.writeFileSync(path.join(target, `${config.circuitName}.pw`), pw);
}

function generateProof(config, inputs) {
  const row = withdrawRow(inputs);
  fs.writeFileSync(path.join(config.circuitDir, "Prover.toml"), proverToml(inputs));
  const r = addon.proveBatch(circuitHandle(config), 1, row, null);
  if (r.status[0] !== 0) throw new Error(`libspp error ${r.status[0]}: inputs do not satisfy the circuit`);
  writeOutputs(config, r.proofs, r.publicWitnesses);
  return { proof: r.proofs, publicWitness: r.publicWitnesses };
}

// Many withdraw proofs in ONE call -- what client/payroll-demo.ts:326-352 wants from its Promise.all over generateProof
// (which, in the reference, serialises on execSync and races on the shared Prover.toml).  Returns one {proof, publicWitness}
// per input, in order; throws naming the first index whose inputs do not satisfy the circuit.  No files are written.
function generateProofBatch(config, inputsList) {
  if (!Array.isArray(inputsList)) throw new Error("generateProofBatch expects an array of ShieldedPoolInputs");
  if (inputsList.length === 0) return [];
  const rows = Buffer.concat(inputsList.map(withdrawRow));
  const r = addon.proveBatch(circuitHandle(config), inputsList.length, rows, null);
  const bad = r.status.findIndex((s) => s !== 0);
  if (bad >= 0) throw new Error(`libspp error ${r.status[bad]}: inputs of proof ${bad} do not satisfy the circuit`);
  const pwLen = r.publicWitnesses.length / inputsList.length;
  return inputsList.map((_, i) => ({ proof: r.proofs.slice(388 * i, 388 * (i + 1)), publicWitness: r.publicWitnesses.slice(pwLen * i, pwLen * (i + 1)) }));
}

// ---- audit circuit: the reference proves it from scripts (audit_circuit/prove_audit.sh:74-99, scripts/generate_audit.py:668-685)
const BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const AUDIT_ORDER = ["secret_key", "wa_commitment", "ct_commitment", "c0_packed", "c1_packed", "r", "e1_sparse", "e2", "k0", "k1"];
const AUDIT_LEN = { c0_packed: 10, c1_packed: 147, r: 1024, e1_sparse: 64, e2: 1024, k0: 64, k1: 1024 };

function toFieldSigned(v) {           // format_field (generate_audit.py:77-82): negative values are stored as p - |v|
  let n = typeof v === "bigint" ? v : BigInt(v);
  n %= BN254_R;
  if (n < 0n) n += BN254_R;
  return n;
}
function formatField(v) {
  const n = toFieldSigned(v);
  return n === 0n ? '"0"' : `"0x${n.toString(16).padStart(64, "0")}"`;
}
// text of scripts/generate_audit.py:630-641 (same key order, single-line arrays)
function auditProverToml(inputs) {
  let out = "";
  for (const k of AUDIT_ORDER) out += Array.isArray(inputs[k]) ? `${k} = [${inputs[k].map(formatField).join(", ")}]\n` : `${k} = ${formatField(inputs[k])}\n`;
  return out;
}
// generateAuditProof(config, auditInputs): auditInputs carries the audit Prover.toml keys (generate_audit.py:630-641):
// secret_key, wa_commitment, ct_commitment, c0_packed[10], c1_packed[147], r[1024], e1_sparse[64], e2[1024], k0[64], k1[1024]
// (hex strings, decimal strings, numbers or bigints; signed values allowed).  Returns { proof: 388 B, publicWitness: 76 B }
// and leaves Prover.toml, target/<name>.proof and .pw behind like generateProof.
function generateAuditProof(config, inputs) {
  for (const [k, n] of Object.entries(AUDIT_LEN)) if (!Array.isArray(inputs[k]) || inputs[k].length !== n) throw new Error(`${k} must hold ${n} elements`);
  const order = ["wa_commitment", "ct_commitment", "c0_packed", "c1_packed", "secret_key", "r", "e1_sparse", "e2", "k0", "k1"];   // main() parameter order :405-417
  const parts = [];
  for (const k of order) for (const v of Array.isArray(inputs[k]) ? inputs[k] : [inputs[k]]) parts.push(toField32(toFieldSigned(v)));
  fs.writeFileSync(path.join(config.circuitDir, "Prover.toml"), auditProverToml(inputs));
  const r = addon.proveBatch(circuitHandle(config), 1, Buffer.concat(parts), null);
  if (r.status[0] !== 0) throw new Error(`libspp error ${r.status[0]}: inputs do not satisfy the circuit`);
  writeOutputs(config, r.proofs, r.publicWitnesses);
  return { proof: r.proofs, publicWitness: r.publicWitnesses };
}

module.exports = { generateProof, generateProofBatch, generateAuditProof, proverToml, auditProverToml, toField32, addon };
