// Run-time specialisation of the narrow-chain kernels (chain_spec.inc) to a chain's shapes: hiprtc compiles the kernel text
// that is embedded in the library (build.py: build/chain_spec_src.h) with the layer table as macros, once per distinct chain
// and process.  Nothing here is a fallback for a missing GPU path: when hiprtc is unusable the GENERIC hand-written kernels of
// kernels_chain_grid.hip run (same bits), and si_chain_kernel_info reports which ones did.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "si_internal.h"

namespace si {

struct SpecKernels {
  hipModule_t mod = nullptr;
  hipFunction_t fused = nullptr;   // si_spec_fused_kernel
  hipFunction_t stack = nullptr;   // si_spec_stack_kernel: activations in registers, the weights in LDS (nullptr: they do not fit)
  hipFunction_t grid = nullptr;    // si_spec_grid_kernel (nullptr: compiled without the loop)
  hipFunction_t perm = nullptr;    // si_spec_perm_kernel
  int nb = 0, m = 0, preg = 0;
  int lds_doubles = 0;             // dynamic LDS of the forward (doubles); the loop adds its own behind it
  int fo_total = 0;                // doubles per weight vector in fragment order
  int wvec_doubles = 0;            // dynamic LDS of the stacked kernel (doubles): the chain's whole weight vector
  int max_wn = 0;                  // max over the matrix layers of in * out (grid of the perm kernel)
  std::string error;               // non-empty: compilation failed (kept so that it is not retried)
};

// the kernels for this chain (layers: L Dense layers, the last one a narrow head), batch sub-tiles NB, head slot SF; M > 0 adds
// the persistent loop for M columns of P with (preg) the workgroup's rows of P in registers.  Process-wide cache; thread-safe.
// Returns nullptr (and why in *err) when the chain is outside the class or hiprtc fails.
const SpecKernels* spec_kernels(const si_layer* layers, int L, int NB, int SF, int M, bool preg, std::string* err);
// is this chain of the class chain_spec.inc handles (and is its preload small enough)?  No compilation.
bool spec_applies(const si_layer* layers, int L, int NB);

}  // namespace si
