// The dataflow sweep kernels (csrc/hip/gs_flow.hpp) and their relayed form (csrc/hip/gs_relay.hpp) for both value types, and
// nothing else: what tools/flow_asm_audit.py and tools/flow_asm_linear.py compile to assembly (tests/test_flow_asm.py).
#include "../algebraicmultigrid.jl_amd/csrc/hip/gs_relay.hpp"
namespace amgh { namespace bw {
template hipError_t sweep_flow<double>(const FlowArgs<double>&, int, size_t, bool, bool, hipStream_t, int, int);
template hipError_t sweep_flow<float>(const FlowArgs<float>&, int, size_t, bool, bool, hipStream_t, int, int);
template hipError_t sweep_relay<double>(const FlowArgs<double>&, int, size_t, bool, bool, hipStream_t, int);
template hipError_t sweep_relay<float>(const FlowArgs<float>&, int, size_t, bool, bool, hipStream_t, int);
} }
