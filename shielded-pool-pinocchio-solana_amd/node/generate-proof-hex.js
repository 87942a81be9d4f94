#!/usr/bin/env node
// Twin of the reference CLI client/generate-proof-hex.ts: reads the four proof / public-witness files and prints
// them as 0x-hex with the same banners (stdout is identical for identical files).  Usage:
//   node generate-proof-hex.js [repoRoot]      (default: the parent directory, as in the reference layout)
"use strict";
const fs = require("fs");
const path = require("path");

const root = process.argv[2] || path.join(__dirname, "..");
const FILES = [
  { label: "Withdraw", kind: "proof", dir: "noir_circuit", base: "shielded_pool_verifier", ext: "proof", title: "1. WITHDRAW PROOF (hex):" },
  { label: "Withdraw", kind: "witness", dir: "noir_circuit", base: "shielded_pool_verifier", ext: "pw", title: "2. WITHDRAW PUBLIC WITNESS (hex):" },
  { label: "Audit", kind: "proof", dir: "audit_circuit", base: "rlwe_audit", ext: "proof", title: "3. AUDIT PROOF (hex):" },
  { label: "Audit", kind: "witness", dir: "audit_circuit", base: "rlwe_audit", ext: "pw", title: "4. AUDIT PUBLIC WITNESS (hex):" },
];
const HINTS = { Withdraw: ["noir_circuit", "shielded_pool_verifier"], Audit: ["audit_circuit", "rlwe_audit"] };
const bar = "=".repeat(60);

function run() {
  console.log(bar);
  console.log("Shielded Pool - Proof to Hex Converter");
  console.log(bar);
  console.log();
  for (const f of FILES) {
    f.path = path.join(root, f.dir, "target", `${f.base}.${f.ext}`);
    if (!fs.existsSync(f.path)) {
      console.error(`Error: ${f.label} ${f.kind} file not found at ${f.path}`);
      if (f.kind === "proof") {
        console.error("\nMake sure you have run:");
        console.error(`  cd ${HINTS[f.label][0]}`);
        console.error("  nargo execute");
        console.error(`  sunspot prove target/${HINTS[f.label][1]}.json ...`);
      }
      process.exit(1);
    }
  }
  for (const f of FILES) f.bytes = fs.readFileSync(f.path);
  FILES.forEach((f, i) => {
    console.log(`${f.label} ${f.kind} file: ${f.path}`);
    console.log(`${f.label} ${f.kind} size: ${f.bytes.length} bytes`);
    if (i % 2 === 1) console.log();
  });
  for (const f of FILES) {
    console.log(bar);
    console.log(f.title);
    console.log(bar);
    console.log();
    console.log("0x" + f.bytes.toString("hex"));
    console.log();
  }
  console.log(bar);
  console.log("Instructions:");
  console.log(bar);
  const fields = ["'Proof (hex)'", "'Public Witness (hex)'", "'Audit Proof (hex)'", "'Audit Public Witness (hex)'"];
  FILES.forEach((f, i) => console.log(`${i + 1}. Copy ${f.title.slice(3, -7)} hex -> paste into ${fields[i]} field`));
  console.log("5. Verify the recipient address matches the one used in Prover.toml");
  console.log("6. Click 'Submit via Relayer'");
  console.log();
}
run();
