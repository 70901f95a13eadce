#include "../blst_eip2537_amd/csrc/pairing.hip"
using namespace eip;
namespace eip { hipError_t DevBuf::reserve(size_t) { return hipSuccess; } void DevBuf::release() {} }
int main() {
    const int k = 2;
    Aff<Fp2> g2{Fp2{Fp{{K_G2_X_C0}}, Fp{{K_G2_X_C1}}}, Fp2{Fp{{K_G2_Y_C0}}, Fp{{K_G2_Y_C1}}}};
    uint32_t rec[96 * k];
    memset(rec, 0, sizeof rec);
    for (int i = 0; i < k; i++) encode_point<Fp2>(rec + 96 * i + 32, g2);
    uint32_t *d_in; unsigned long long *d_err, herr;
    hipMalloc(&d_in, sizeof rec); hipMalloc(&d_err, 8);
    hipMemcpy(d_in, rec, sizeof rec, hipMemcpyHostToDevice);
    hipMemset(d_err, 0xFF, 8);
    k_pair_check_g2<<<1, 64>>>(d_in, k, d_err);
    hipMemcpy(&herr, d_err, 8, hipMemcpyDeviceToHost);
    printf("g2 check err word = %llx (ffffffffffffffff = no error)\n", herr);
    hipMemset(d_err, 0xFF, 8);
    k_pair_check_g1<<<1, 64>>>(d_in, k, d_err);
    hipMemcpy(&herr, d_err, 8, hipMemcpyDeviceToHost);
    printf("g1 check err word = %llx\n", herr);
    return 0;
}
