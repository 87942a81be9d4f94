// Stand-alone witness-input kernels (RLWE negacyclic, Poseidon-Merkle, Grumpkin keygen) -- added below.
#include "kernels.hpp"
namespace spp {}
