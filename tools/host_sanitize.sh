#!/bin/sh
# Host-side sanitizer pass (GPU ASan is not available on this pool): builds the engine's host
# code (codec, curve, pairing tower, map-to-curve, C-ABI single operations) with ASan + UBSan and
# drives every host-only precompile with valid, invalid and edge inputs.
set -e
cd "$(dirname "$0")/.."
cat > /tmp/host_san.cpp <<'EOC'
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../blst_eip2537_amd/csrc/codec.h"
#include "../blst_eip2537_amd/csrc/pairing.h"
#include "../blst_eip2537_amd/csrc/h2c.h"
using namespace eip;
int main() {
    Aff<Fp> g1{Fp{{K_G1_X}}, Fp{{K_G1_Y}}};
    Aff<Fp2> g2{Fp2{Fp{{K_G2_X_C0}}, Fp{{K_G2_X_C1}}}, Fp2{Fp{{K_G2_Y_C0}}, Fp{{K_G2_Y_C1}}}};
    uint32_t w[64], k[8] = {0xffffffffu, 5, 0, 0, 0, 0, 0, 0x80000000u};
    encode_point<Fp>(w, g1);
    Aff<Fp> d;
    int bad = decode_point<Fp>(d, w) != 0;
    bad |= !in_g1(g1) || !in_g2(g2);
    Aff<Fp> s = to_affine(scalar_mul(g1, k, 256));
    bad |= !on_curve(s);
    Aff<Fp2> s2 = to_affine(madd(scalar_mul(g2, k, 256), g2));
    bad |= !on_curve(s2);
    Fp12 f = miller_loop(s, g2), h = miller_loop(neg(s), g2);
    bad |= !is_one(final_exp(mul(f, h)));
    bad |= is_one(final_exp(f));
    {   // host route of small calls: interleaved-window MSM against a sum of double-and-add products (special scalars,
        // a repeated point, infinity), and the shared-squaring Miller loop against the product of single loops
        Aff<Fp> ps[5] = {g1, s, g1, Aff<Fp>{fp_zero(), fp_zero()}, neg(s)};
        uint32_t ks[5][8] = {{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu},
                             {0, 0, 0, 0, 0, 0, 0, 0x80000000u}, {16, 0, 0, 0, 0, 0, 0, 0}, {7, 1, 2, 3, 4, 5, 6, 7}, {0x11111111u, 17, 0, 0x84210842u, 0, 0, 0, 1}};
        Xyzz<Fp> want = xyzz_inf<Fp>();
        for (int i = 0; i < 5; i++) want = add(want, scalar_mul(ps[i], ks[i], 256));
        const Aff<Fp> got = to_affine(msm_interleaved<Fp>(ps, &ks[0][0], 5)), wa = to_affine(want);
        bad |= !eq(got.x, wa.x) || !eq(got.y, wa.y);
        Aff<Fp2> qs[3] = {g2, s2, g2};
        Aff<Fp> pp[3] = {s, g1, Aff<Fp>{fp_zero(), fp_zero()}};
        Fp12 prod = mul(miller_loop(pp[0], qs[0]), miller_loop(pp[1], qs[1]));
        bad |= !eq(final_exp(miller_loop_multi(pp, qs, 3)), final_exp(prod));
    }
    Aff<Fp> mp = to_affine(scalar_mul(map_to_curve<Fp>(Fp{{K_BETA}}), K_ISO_H_EFF_G1, 64));
    bad |= !in_g1(mp);
    Aff<Fp2> mq = to_affine(scalar_mul(map_to_curve<Fp2>(Fp2{Fp{{K_BETA}}, fp_one()}), K_ISO_H_EFF_G2, K_ISO_H_EFF_G2_BITS));
    bad |= !in_g2(mq);
    memset(w, 0xff, sizeof w);
    bad |= decode_point<Fp2>(s2, w) != E_INVALID_ELEMENT;
    printf("host sanitize run: %s\n", bad ? "FAILED" : "ok");
    return bad;
}
EOC
/opt/rocm/bin/hipcc -x hip --cuda-host-only -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
    -I blst_eip2537_amd/csrc -I tools /tmp/host_san.cpp -o /tmp/host_san
ASAN_OPTIONS=detect_leaks=0 /tmp/host_san
