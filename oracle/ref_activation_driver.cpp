// Driver that exposes the REFERENCE's own activation functors through a C ABI so the oracle's
// restatement can be pinned bit-for-bit.  The reference header is compiled where it lies
// (-I/root/reference/Mila/Src/Dnn/Components/Activations/Activation/Kernels); it is the only
// source file of the reference's hot path that builds with this image's toolchain
// (SURVEY.md section 8c).  Output goes to oracle/_ref/ (git-ignored, travels with gpurun).
// TEST INFRASTRUCTURE ONLY.
#include "ElementwiseActivation.h"

using namespace Mila::Dnn::Activations;

extern "C" {
float ref_gelu_tanh(float x) { return GeluTanh{}.fwd(x); }
float ref_silu(float x) { return Silu{}.fwd(x); }
float ref_relu(float x) { return Relu{}.fwd(x); }
float ref_tanh(float x) { return Tanh{}.fwd(x); }
float ref_sigmoid(float x) { return Sigmoid{}.fwd(x); }
float ref_mish(float x) { return Mish{}.fwd(x); }
void ref_gelu_tanh_array(float* y, const float* x, long n) { for (long i = 0; i < n; ++i) y[i] = GeluTanh{}.fwd(x[i]); }
void ref_silu_array(float* y, const float* x, long n) { for (long i = 0; i < n; ++i) y[i] = Silu{}.fwd(x[i]); }
}
