// Test-only host driver: build a circuit with the product's builder and write the SPPC file.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "circuit.hpp"
#include <vector>
using namespace spp;
int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: build_circuit withdraw|withdraw-generic out.sppc\n"); return 2; }
  Circuit c;
  if (!strcmp(argv[1], "withdraw")) c = build_withdraw_circuit(true);
  else if (!strcmp(argv[1], "withdraw-generic")) c = build_withdraw_circuit(false);
  else if (!strcmp(argv[1], "withdraw-refshape")) c = build_withdraw_circuit(true, 12452);
  else if (!strcmp(argv[1], "withdraw-depth20")) c = build_withdraw_circuit(true, 0, 20);
  else if (!strcmp(argv[1], "audit") || !strcmp(argv[1], "audit-generic")) {
    // argv[3]: text file with 2048 integers (a then b)
    std::vector<uint32_t> pk;
    FILE* f = fopen(argv[3], "r");
    unsigned v;
    while (f && fscanf(f, "%u", &v) == 1) pk.push_back(v);
    if (pk.size() != 2048) { fprintf(stderr, "need 2048 pk coefficients\n"); return 2; }
    c = build_audit_circuit(pk.data(), pk.data() + 1024, !strcmp(argv[1], "audit"));
  }
  else return 2;
  printf("circuit %u: public=%u secret=%u wires=%u constraints=%u domain=2^%u nnzA=%zu nnzB=%zu nnzC=%zu hrows=%u committed=%zu prog=%zu coeffs=%zu\n",
         c.id, c.n_public, c.n_secret, c.n_wires, c.n_constraints, c.domain_log, c.A.terms.size(), c.B.terms.size(),
         c.C.terms.size(), c.H.rows(), c.committed.size(), c.program.size(), c.coeffs.size());
  return c.save(argv[2]) ? 0 : 1;
}
