// stft4096.hip -- tuned 4096-point STFT (placeholder until the wave-per-frame kernel lands)
#include "sgx_internal.hpp"
namespace sgx {
bool fast4096_supported(const sgx_ctx *) { return false; }
hipError_t fast4096_init(sgx_ctx *) { return hipSuccess; }
void fast4096_destroy(sgx_ctx *) {}
hipError_t launch_stft_fast4096(const sgx_ctx *, const float *, uint32_t, uint32_t, size_t, size_t, float *) { return hipErrorNotSupported; }
}
