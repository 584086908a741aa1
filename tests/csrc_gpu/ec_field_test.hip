// The two forms of the secp256k1 base-field arithmetic of the accumulator chain (era_zkevm_test_harness_amd/csrc/ec_field.cuh: ecf, a value
// in one lane; ecl, a limb per lane) against the host arithmetic of include/zkw_ecrecover.h (ec_mulmod / ec_addmod / ec_submod), on edge
// values (0, 1, p - 1, values whose words are all ones or end in long runs of ones: the carries that ripple) and seeded random ones; and
// the Jacobian doubling / mixed addition of both forms against each other. Prints "ok <cases>" or the first mismatch. Test infrastructure
// (tests/test_gpu_ec_field.py builds and runs it on the GPU box).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../era_zkevm_test_harness_amd/csrc/ec_field.cuh"

using namespace zkw;

struct Out { ec_u256 mul_l, add_l, sub_l, mul_f, add_f, sub_f, jd_l[3], jd_f[3], ja_l[3], ja_f[3], sqrt_l; };

__device__ ec_u256 gather(u32 v) {  // the limbs of a lane-form value, in lane 0
    ec_u256 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = __builtin_amdgcn_readlane(v, i);
    return r;
}

__global__ void k_test(const ec_u256* A, const ec_u256* B, Out* out, int n) {
    const u32 lane = threadIdx.x;
    if (lane >= 16) return;
    for (int i = 0; i < n; i++) {
        const u32 a = lane < 8 ? A[i].w[lane] : 0, b = lane < 8 ? B[i].w[lane] : 0;
        const u32 m = ecl::mul(a, b), s = ecl::add(a, b), d = ecl::sub(a, b);
        u32 X = a, Y = b, Z = m;
        ecl::jdbl(X, Y, Z);
        u32 X2 = a, Y2 = b, Z2 = m;
        ecl::jmadd(X2, Y2, Z2, s, d);
        const ec_u256 gq = gather(i % 37 == 0 ? ecl::pow_sqrt(a) : 0u);  // (a^((p + 1) / 4): ~500 multiplications, on a sample of the pairs)
        const ec_u256 gm = gather(m), gs = gather(s), gd = gather(d), gx = gather(X), gy = gather(Y), gz = gather(Z), hx = gather(X2), hy = gather(Y2), hz = gather(Z2);
        if (lane == 0) {
            Out& o = out[i];
            o.mul_l = gm; o.add_l = gs; o.sub_l = gd; o.sqrt_l = gq;
            o.jd_l[0] = gx; o.jd_l[1] = gy; o.jd_l[2] = gz;
            o.ja_l[0] = hx; o.ja_l[1] = hy; o.ja_l[2] = hz;
            const ec_u256 fa = A[i], fb = B[i];
            o.mul_f = ecf::mul(fa, fb); o.add_f = ecf::add(fa, fb); o.sub_f = ecf::sub(fa, fb);
            ec_u256 x = fa, y = fb, z = o.mul_f;
            ecf::jdbl(x, y, z);
            o.jd_f[0] = x; o.jd_f[1] = y; o.jd_f[2] = z;
            x = fa; y = fb; z = o.mul_f;
            ecf::jmadd(x, y, z, o.add_f, o.sub_f);
            o.ja_f[0] = x; o.ja_f[1] = y; o.ja_f[2] = z;
        }
    }
}

static bool eq(const ec_u256& a, const ec_u256& b) { return memcmp(a.w, b.w, 32) == 0; }
static void show(const char* what, const ec_u256& v) {
    printf("  %s ", what);
    for (int i = 7; i >= 0; i--) printf("%08x", v.w[i]);
    printf("\n");
}

int main() {
    const ec_mod M = ec_modulus(0);
    std::vector<ec_u256> vals;
    auto from_words = [](std::initializer_list<uint32_t> w) { ec_u256 v; int i = 0; for (uint32_t x : w) v.w[i++] = x; for (; i < 8; i++) v.w[i] = 0; return v; };
    ec_u256 pm1;
    for (int i = 0; i < 8; i++) pm1.w[i] = EC_P_M[i];
    pm1.w[0] -= 1;
    vals.push_back(from_words({0}));
    vals.push_back(from_words({1}));
    vals.push_back(from_words({2}));
    vals.push_back(pm1);
    { ec_u256 v = pm1; v.w[0] -= 1; vals.push_back(v); }
    vals.push_back(from_words({977, 1}));                                                                              // c
    vals.push_back(from_words({976, 1}));
    vals.push_back(from_words({0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu}));
    vals.push_back(from_words({0xFFFFFC2Eu - 977u, 0xFFFFFFFDu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}));  // p - 1 - c
    vals.push_back(from_words({0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu}));
    vals.push_back(from_words({0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0}));
    vals.push_back(from_words({0, 0, 0, 0, 0, 0, 0, 0x80000000u}));
    vals.push_back(from_words({0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}));
    vals.push_back(from_words({0, 0, 0, 0, 1}));
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
    for (int k = 0; k < 160; k++) {
        ec_u256 v;
        for (int i = 0; i < 8; i++) v.w[i] = rnd();
        if (k % 4 == 1) for (int i = 2; i < 7; i++) v.w[i] = 0xFFFFFFFFu;  // runs of ones: carries ripple
        if (k % 4 == 2) for (int i = 1; i < 8; i++) v.w[i] = i < 5 ? 0 : v.w[i];
        if (ec_cmp8(v.w, EC_P_M) >= 0) v.w[7] &= 0x7FFFFFFFu;
        vals.push_back(v);
    }
    std::vector<ec_u256> A, B;
    for (auto& a : vals) for (auto& b : vals) { A.push_back(a); B.push_back(b); }
    const int n = (int)A.size();
    ec_u256 *dA, *dB;
    Out* dO;
    if (hipMalloc(&dA, n * sizeof(ec_u256)) != hipSuccess || hipMalloc(&dB, n * sizeof(ec_u256)) != hipSuccess || hipMalloc(&dO, n * sizeof(Out)) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
    (void)hipMemcpy(dA, A.data(), n * sizeof(ec_u256), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), n * sizeof(ec_u256), hipMemcpyHostToDevice);
    (void)hipMemset(dO, 0xEE, n * sizeof(Out));
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, dO, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    std::vector<Out> O(n);
    (void)hipMemcpy(O.data(), dO, n * sizeof(Out), hipMemcpyDeviceToHost);
    ec_ws W;
    for (int pass = 0; pass < 2; pass++)  // the three operations on every pair first, then the point formulas built from them
        for (int i = 0; i < n; i++) {
            const ec_u256 m = ec_mulmod(&A[i], &B[i], &M, &W), s = ec_addmod(&A[i], &B[i], &M, &W), d = ec_submod(&A[i], &B[i], &M);
            const char* bad = nullptr;
            if (pass == 0) {
                if (!eq(O[i].mul_f, m)) bad = "ecf::mul";
                else if (!eq(O[i].add_f, s)) bad = "ecf::add";
                else if (!eq(O[i].sub_f, d)) bad = "ecf::sub";
                else if (!eq(O[i].mul_l, m)) bad = "ecl::mul";
                else if (!eq(O[i].add_l, s)) bad = "ecl::add";
                else if (!eq(O[i].sub_l, d)) bad = "ecl::sub";
                else if (i % 37 == 0) {
                    const ec_u256 q = ec_powmod(&A[i], EC_P_SQRT_E, &M, &W);
                    if (!eq(O[i].sqrt_l, q)) bad = "ecl::pow_sqrt";
                }
            } else
                for (int k = 0; k < 3 && !bad; k++) {
                    if (!eq(O[i].jd_l[k], O[i].jd_f[k])) bad = "jdbl";
                    else if (!eq(O[i].ja_l[k], O[i].ja_f[k])) bad = "jmadd";
                }
            if (bad) {
                printf("MISMATCH %s at case %d\n", bad, i);
                show("a      ", A[i]); show("b      ", B[i]); show("mul    ", m); show("ecl mul", O[i].mul_l); show("add    ", s); show("ecl add", O[i].add_l); show("sub    ", d); show("ecl sub", O[i].sub_l);
                return 1;
            }
        }
    printf("ok %d\n", n);
    return 0;
}
