// RLWE audit circuit (scripts/generate_audit.py:405-463 of the reference) -- see build_audit_circuit below.
#include "circuit.hpp"
namespace spp {
Circuit build_audit_circuit(const uint32_t* pk_a, const uint32_t* pk_b, bool native_hints) {
  (void)pk_a; (void)pk_b; (void)native_hints;
  return Circuit();
}
}  // namespace spp
