// devtools/gemm_lab.cpp -- measurement harness for the wide-update GEMMs (not product code).
// Build: hipcc -O2 -std=c++17 devtools/gemm_lab.cpp -Icuda-qr_amd/csrc -Lcuda-qr_amd -lmi355xqr -Wl,-rpath,$PWD/cuda-qr_amd -o gpurun_out/gemm_lab
// Checks gemm_nt / gemm_tnt against a host reference on a small problem, then times old and new kernels in
// interleaved rounds (one process, HIP events) on the whole chip and on CU-masked streams.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "qr_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) { printf("rc %d at %s:%d\n", rc_, __FILE__, __LINE__); exit(3); } } while (0)

static double* dalloc(size_t n) { double* p; CK(hipMalloc(&p, n * sizeof(double))); return p; }

static int check_small()
{
    const int M = 384, N = 256, K = 64;
    std::vector<double> A((size_t) M * K), Bt((size_t) N * K), C((size_t) M * N), R, G((size_t) M * N);
    srand(1);
    for (auto& v : A) v = rand() / (double) RAND_MAX - 0.5;
    for (auto& v : Bt) v = rand() / (double) RAND_MAX - 0.5;
    for (auto& v : C) v = rand() / (double) RAND_MAX - 0.5;
    R = C;
    for (int j = 0; j < N; ++j)
        for (int i = 0; i < M; ++i) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += A[(size_t) k * M + i] * Bt[(size_t) k * N + j];
            R[(size_t) j * M + i] -= s;
        }
    double *dA = dalloc(A.size()), *dB = dalloc(Bt.size()), *dC = dalloc(C.size());
    int bad = 0;
    for (int gm : {0, 8, 3}) {
        CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, Bt.data(), Bt.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
        RC(qrd_gemm_nt(nullptr, M, N, K, -1, dA, M, dB, N, dC, M, gm, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(G.data(), dC, G.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, nrm = 0;
        for (size_t e = 0; e < G.size(); ++e) { err = std::max(err, fabs(G[e] - R[e])); nrm = std::max(nrm, fabs(R[e])); }
        printf("check gemm_nt  gm=%d  max|err| = %.3e (max|ref| %.3f)\n", gm, err, nrm);
        if (!(err < 1e-12)) bad = 1;
    }
    // plus sign
    {
        CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
        RC(qrd_gemm_nt(nullptr, M, N, K, +1, dA, M, dB, N, dC, M, 8, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(G.data(), dC, G.size() * 8, hipMemcpyDeviceToHost));
        double err = 0;
        for (size_t e = 0; e < G.size(); ++e) err = std::max(err, fabs(G[e] - (2 * C[e] - R[e])));
        printf("check gemm_nt  sign=+1 max|err| = %.3e\n", err);
        if (!(err < 1e-12)) bad = 1;
    }
    // tnt: Ct (M2 x N2) = P^T Q, P: K2 x M2, Q: K2 x N2
    const int M2 = 256, N2 = 128, K2 = 1040;
    std::vector<double> P((size_t) K2 * M2), Q((size_t) K2 * N2), Rt((size_t) M2 * N2), Gt((size_t) M2 * N2);
    for (auto& v : P) v = rand() / (double) RAND_MAX - 0.5;
    for (auto& v : Q) v = rand() / (double) RAND_MAX - 0.5;
    for (int j = 0; j < N2; ++j)
        for (int i = 0; i < M2; ++i) {
            double s = 0;
            for (int k = 0; k < K2; ++k) s += P[(size_t) i * K2 + k] * Q[(size_t) j * K2 + k];
            Rt[(size_t) j * M2 + i] = s;
        }
    double *dP = dalloc(P.size()), *dQ = dalloc(Q.size()), *dT = dalloc(Rt.size()), *dS = dalloc(8 * Rt.size());
    CK(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dQ, Q.data(), Q.size() * 8, hipMemcpyHostToDevice));
    for (int ks : {1, 3, 0}) {
        CK(hipMemset(dT, 0xff, Rt.size() * 8));
        RC(qrd_gemm_tnt(nullptr, M2, N2, K2, dP, K2, dQ, K2, dT, M2, dS, 8 * Rt.size(), ks, 256, 8));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(Gt.data(), dT, Gt.size() * 8, hipMemcpyDeviceToHost));
        double err = 0;
        for (size_t e = 0; e < Gt.size(); ++e) err = std::max(err, fabs(Gt[e] - Rt[e]));
        printf("check gemm_tnt ksplit=%d max|err| = %.3e\n", ks, err);
        if (!(err < 1e-11)) bad = 1;
    }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dP); hipFree(dQ); hipFree(dT); hipFree(dS);
    return bad;
}

struct Variant { std::string name; std::function<int(hipStream_t)> run; double flops; std::vector<double> ms; };

static void time_variants(hipStream_t s, std::vector<Variant>& vs, int rounds, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& v : vs) { RC(v.run(s)); CK(hipStreamSynchronize(s)); if (getenv("LAB_TRACE")) printf("      warm-up done: %s\n", v.name.c_str()); }
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(a, s));
            for (int i = 0; i < reps; ++i) RC(v.run(s));
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            v.ms.push_back(ms / reps);
        }
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2], mn = v.ms[0];
        printf("    %-34s  median %8.3f ms  %6.2f TFLOP/s   (best %8.3f ms %6.2f)\n", v.name.c_str(), med, v.flops / med / 1e9, mn, v.flops / mn / 1e9);
        v.ms.clear();
    }
    hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    RC(qrd_init());
    if (check_small()) { printf("CORRECTNESS FAILED\n"); return 1; }
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const int MM = 16384;
    double* C = dalloc((size_t) MM * MM);
    double* V = dalloc((size_t) MM * 512);
    double* Wt = dalloc((size_t) MM * 512);
    double* W = dalloc((size_t) MM * 512);
    const size_t slab_cap = (size_t) 16 << 20;
    double* slabs = dalloc(slab_cap);
    RC(qrd_fill_uniform(nullptr, C, MM, MM, MM, 0, MM, 1));
    RC(qrd_fill_uniform(nullptr, V, MM, MM, 512, 0, MM, 2));
    RC(qrd_fill_uniform(nullptr, Wt, MM, MM, 512, 0, MM, 3));
    RC(qrd_fill_uniform(nullptr, W, 512, 512, MM, 0, 512, 4));
    CK(hipDeviceSynchronize());

    struct StreamSpec { const char* name; int first, count; };
    std::vector<StreamSpec> specs = {{"whole chip (256 CUs)", 0, 0}, {"CU mask 192 (64..255)", 64, 192}, {"CU mask 224 (32..255)", 32, 224},
                                     {"CU mask 240 (16..255)", 16, 240}};
    if (quick) specs.resize(2);
    if (argc > 1 && !strcmp(argv[1], "m224")) specs = {{"CU mask 224 (32..255)", 32, 224}};
    struct Shape { int M, N, K; };
    std::vector<Shape> shapes = {{16384, 16128, 256}, {8192, 7936, 256}, {4096, 3840, 256}, {16384, 15872, 512}, {16384, 16256, 128}};
    if (quick) shapes.resize(2);
    for (auto& sp : specs) {
        void* sv = nullptr;
        RC(qrd_stream_create_cumask(&sv, sp.first, sp.count));
        hipStream_t s = (hipStream_t) sv;
        const int cus = sp.count ? sp.count : 256;
        printf("== stream: %s\n", sp.name);
        for (auto& sh : shapes) {
            const int M = sh.M, N = sh.N, K = sh.K;
            const double fl = 2.0 * M * (double) N * K;
            printf("  update A2 -= V W : M=%d N=%d K=%d  (%.2f GFLOP)\n", M, N, K, fl / 1e9);
            std::vector<Variant> vs;
            vs.push_back({"old gemm_nn_w8", [=](hipStream_t st) { return qrd_gemm_nn_update(st, M, N, K, -1.0, V, MM, W, K, 1.0, C, MM); }, fl, {}});
            for (int gm : {0, 4, 8, 16})
                vs.push_back({"new gemm_nt gm=" + std::to_string(gm), [=](hipStream_t st) { return qrd_gemm_nt(st, M, N, K, -1, V, MM, Wt, MM, C, MM, gm, nullptr); }, fl, {}});
            time_variants(s, vs, 3, 3);
            printf("  product W = (VT)^T A2 : K(long)=%d nt=%d nb=%d\n", M, N, K);
            std::vector<Variant> vt;
            vt.push_back({"old gemm_tn<4,4> + slab_reduce", [=](hipStream_t st) { return qrd_gemm_tn_update(st, K, N, M, 1.0, V, MM, C, MM, 0.0, W, K, slabs, slab_cap); }, fl, {}});
            if (K % 128 == 0)
                for (int gm : {0, 8})
                    vt.push_back({"new gemm_tnt gm=" + std::to_string(gm), [=](hipStream_t st) { return qrd_gemm_tnt(st, N, K, M, C, MM, V, MM, Wt, N, slabs, slab_cap, 0, cus, gm); }, fl, {}});
            time_variants(s, vt, 3, 3);
        }
        RC(qrd_stream_destroy(sv));
    }
    // phase breakdown of gemm_nt from in-kernel s_memtime stamps (whole chip, biggest shape), and how the two workgroups that share a
    // compute unit sit relative to each other in time
    {
        const int M = 16384, N = 16128, K = 256, nwg = (M / 128) * (N / 128);
        unsigned long long* st; CK(hipMalloc(&st, sizeof(unsigned long long) * 6 * nwg));
        RC(qrd_gemm_nt(nullptr, M, N, K, -1, V, MM, Wt, MM, C, MM, 8, st));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(6 * (size_t) nwg);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        double p = 0, l = 0, e = 0;
        for (int i = 0; i < nwg; ++i) { p += h[6 * i + 1] - h[6 * i]; l += h[6 * i + 2] - h[6 * i + 1]; e += h[6 * i + 3] - h[6 * i + 2]; }
        printf("gemm_nt stamps (s_memtime ticks, mean per workgroup): prologue %.0f  main loop %.0f  epilogue %.0f   [K=256: 16 k-steps]\n", p / nwg, l / nwg, e / nwg);
        // per compute unit (xcc, se, sh, cu of HW_ID): time with 2 / 1 / 0 workgroups inside their K loop, between the first loop start
        // and the last loop end on that CU
        struct Iv { unsigned long long a, b; };
        std::vector<std::vector<Iv>> cu(8 * 64 * 4);
        for (int i = 0; i < nwg; ++i) {
            const unsigned long long hw = h[6 * i + 4];
            const unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = (unsigned) (hw >> 32) & 0xf;
            cu[((xcc * 8 + se) * 2 + sh) * 16 + cuid].push_back({h[6 * i + 1], h[6 * i + 2]});
        }
        double t2 = 0, t1 = 0, t0 = 0; int ncu = 0; size_t maxwg = 0, minwg = 1 << 30;
        for (auto& v : cu) {
            if (v.empty()) continue;
            ++ncu; maxwg = std::max(maxwg, v.size()); minwg = std::min(minwg, v.size());
            std::vector<std::pair<unsigned long long, int>> ev;
            for (auto& iv : v) { ev.push_back({iv.a, +1}); ev.push_back({iv.b, -1}); }
            std::sort(ev.begin(), ev.end());
            int depth = 0; unsigned long long last = ev[0].first;
            for (auto& x : ev) {
                const double d = (double) (x.first - last);
                if (depth >= 2) t2 += d; else if (depth == 1) t1 += d; else t0 += d;
                depth += x.second; last = x.first;
            }
        }
        // progress rates: a workgroup's K loop needs 65536 matrix-pipe cycles per SIMD (16 k-tiles x 32 MFMAs x 64 cycles x its 2 waves
        // per SIMD); least squares of 65536 = a * T_shared + b * T_solo over all workgroups (T in s_memtime ticks = shader cycles):
        // a = share of the pipe a workgroup gets while its partner is in its K loop too, b = while it is alone
        {
            double Sss = 0, Sso = 0, Soo = 0, Bs = 0, Bo = 0;
            for (auto& v : cu) {
                for (size_t i = 0; i < v.size(); ++i) {
                    double tsh = 0;
                    for (size_t j = 0; j < v.size(); ++j) {
                        if (j == i) continue;
                        const unsigned long long lo = std::max(v[i].a, v[j].a), hi = std::min(v[i].b, v[j].b);
                        if (hi > lo) tsh += (double) (hi - lo);
                    }
                    const double tso = (double) (v[i].b - v[i].a) - tsh;
                    Sss += tsh * tsh; Sso += tsh * tso; Soo += tso * tso; Bs += 65536.0 * tsh; Bo += 65536.0 * tso;
                }
            }
            const double det = Sss * Soo - Sso * Sso;
            if (det != 0.0)
                printf("gemm_nt K-loop progress: %.3f of the matrix pipe per workgroup while both are in their K loop (%.3f together), %.3f while alone\n",
                       (Bs * Soo - Bo * Sso) / det, 2 * (Bs * Soo - Bo * Sso) / det, (Bo * Sss - Bs * Sso) / det);
        }
        const double tt = t2 + t1 + t0;
        printf("gemm_nt K-loop overlap per compute unit (%d CUs seen, %zu..%zu workgroups each): both workgroups in their K loop %.1f %%, one %.1f %%, none %.1f %% of the time\n",
               ncu, minwg, maxwg, 100 * t2 / tt, 100 * t1 / tt, 100 * t0 / tt);
    }
    return 0;
}
