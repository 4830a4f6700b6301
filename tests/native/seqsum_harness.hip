// Host-side check of jlama_amd/csrc/jh_seqsum.h: the block-parallel evaluation of a float running sum (the scheme of
// sample_pick_kernel, jh_kernels.h) against the plain loop of AbstractModel.sample (AbstractModel.java:475-489), on inputs that
// exercise binade crossings, ties, denormals and zeros.  Test infrastructure: built and run by tests/test_seqsum.py (no GPU).
#include "../../jlama_amd/csrc/jh_seqsum.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace jh;

struct Result { float s; int pick; long iterations; };
// plain: s += v[i] / div (div == 0: no division); stops at the first s >= u
static Result plain(const std::vector<float>& v, float u, float div) {
    float s = 0.0f;
    for (size_t i = 0; i < v.size(); i++) {
        s += div != 0.0f ? v[i] / div : v[i];
        if (s >= u) return {s, (int)i, 0};
    }
    return {s, -1, 0};
}
// the kernel's scheme with T lanes of E elements per iteration and a walker of W elements
static Result blocked(const std::vector<float>& v, float u, float div, int T, int E, int W) {
    const int V = (int)v.size();
    auto val = [&](int e) { return e < V ? (div != 0.0f ? v[e] / div : v[e]) : 0.0f; };
    float s = 0.0f;
    int pos = 0;
    long it = 0;
    std::vector<int> M(T);
    while (pos < V) {
        it++;
        const int k = seq_k(s), m_in = seq_m(s), th = seq_threshold(u, k);
        SeqStep pre{0, 0};
        int te = -1;
        for (int t = 0; t < T; t++) {                       // (a scan in the kernel; composition is associative)
            SeqRef ref = seq_ref_begin(k);
            for (int i = 0; i < E; i++) seq_ref_add(ref, val(pos + t * E + i));
            pre = seq_compose(pre, seq_ref_end(ref, k));
            M[t] = seq_apply(m_in, pre);
            if (te < 0 && M[t] >= th) te = t;
        }
        if (te < 0) { s = seq_value(M[T - 1], k); pos += T * E; continue; }
        s = seq_value(te == 0 ? m_in : M[te - 1], k);
        const int e0 = pos + te * E;
        for (int i = 0; i < W && e0 + i < V; i++) {   // (W = E in the kernel: the lane's own elements)
            s += val(e0 + i);
            if (s >= u) return {s, e0 + i, it};
        }
        pos = e0 + W;
    }
    return {s, -1, it};
}
static int failures = 0;
static long worst_iterations = 0, worst_real = 0;
static void check(const char* what, const std::vector<float>& v, float u, float div, int T, int E, int W) {
    const Result a = plain(v, u, div), b = blocked(v, u, div, T, E, W);
    if (b.iterations > worst_iterations) worst_iterations = b.iterations;
    if (T == 1024 && E == 16 && b.iterations > worst_real) worst_real = b.iterations;
    const bool ok = a.pick == b.pick && (a.pick >= 0 || seq_bits(a.s) == seq_bits(b.s));
    if (!ok) {
        failures++;
        printf("MISMATCH %s V=%zu u=%g div=%g T=%d E=%d: plain (s=%a pick=%d) blocked (s=%a pick=%d)\n", what, v.size(), u, div, T, E, a.s, a.pick, b.s, b.pick);
    }
}
int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U01(0.0, 1.0);
    std::normal_distribution<double> N01(0.0, 1.0);
    const float inf = INFINITY;
    // one element: the map of the reference runs against the float add, both parities, random binades
    for (int rep = 0; rep < 2000000; rep++) {
        const int k = (int)(rng() % 60) - 30;
        const unsigned m = 0x800000u | (unsigned)(rng() & 0x7fffff);
        const float s = seq_value((int)m, k);
        const int ex = k - (int)(rng() % 30) + 2;
        float x = std::ldexp((float)(0x800000u | (unsigned)(rng() & 0x7fffff)), ex - 23);
        if (rep % 7 == 0) x = std::ldexp(1.0f, ex);                       // powers of two: ties
        if (rep % 11 == 0) x = std::ldexp((float)(rng() & 0xff), ex - 8);  // few significant bits: more ties
        SeqRef ref = seq_ref_begin(k);
        seq_ref_add(ref, x);
        const int after = seq_apply((int)m, seq_ref_end(ref, k));
        const float want = s + x;
        if (after < SEQ_LIMIT) {
            if (seq_bits(seq_value(after, k)) != seq_bits(want)) { failures++; if (failures < 10) printf("step mismatch s=%a x=%a want %a got %a\n", s, x, want, seq_value(after, k)); }
        } else if (want < std::ldexp(1.0f, k + 1)) { failures++; if (failures < 10) printf("false crossing s=%a x=%a\n", s, x); }
    }
    for (int rep = 0; rep < 1000000; rep++) {               // k = -126: denormal and first normal binade, M < 2^24
        const unsigned m = (unsigned)(rng() & 0xffffff);
        const float s = seq_value((int)m, -126);
        float x = seq_float((unsigned)(rng() & 0x1ffffff));  // up to 2^-124
        if (rep % 5 == 0) x = seq_float((unsigned)(rng() & 0xfff));
        SeqRef ref = seq_ref_begin(-126);
        seq_ref_add(ref, x);
        const int after = seq_apply((int)m, seq_ref_end(ref, -126));
        const float want = s + x;
        if (after < SEQ_LIMIT) {
            if (seq_bits(seq_value(after, -126)) != seq_bits(want)) { failures++; if (failures < 10) printf("denormal step mismatch s=%a x=%a want %a got %a\n", s, x, want, seq_value(after, -126)); }
        } else if (want < std::ldexp(1.0f, -125) && x < std::ldexp(1.0f, -126)) {   // (x >= 2^-126 on a denormal s is handed to the plain walk: conservative)
            failures++; if (failures < 10) printf("false crossing (denormal) s=%a x=%a\n", s, x);
        }
    }
    // softmax-shaped inputs at several temperatures and vocabulary sizes (sample_exp_kernel's values)
    for (int V : {1, 7, 8, 1000, 8192, 8193, 50257, 128256}) {
        for (double T : {0.02, 0.1, 0.3, 0.8, 1.5, 10.0}) {
            for (int rep = 0; rep < (V > 20000 ? 3 : 20); rep++) {
                std::vector<double> l(V);
                double mx = -1e300;
                for (auto& x : l) { x = 3.0 * N01(rng) + (U01(rng) < 0.01 ? 8.0 : 0.0); mx = std::max(mx, x); }
                std::vector<float> v(V);
                for (int i = 0; i < V; i++) v[i] = (float)std::exp((l[i] - mx) / T);
                check("softmax sum", v, inf, 0.0f, 1024, 16, 16);
                check("softmax sum small block", v, inf, 0.0f, 64, 4, 4);
                const float sum = plain(v, inf, 0.0f).s;
                for (float u : {0.0f, 1e-9f, 0.01f, 0.37f, 0.5f, 0.93f, 0.999999f, 1.0f, 1.5f, (float)U01(rng)}) {
                    check("softmax pick", v, u, sum, 1024, 16, 16);
                    check("softmax pick small block", v, u, sum, 32, 8, 8);
                }
            }
        }
    }
    // equal logits (every add is the same value: a crossing every doubling), wide exponent ranges, denormals, zeros, tie-heavy values
    for (int V : {4096, 128256}) {
        check("all ones", std::vector<float>(V, 1.0f), inf, 0.0f, 1024, 16, 16);
        check("all ones pick", std::vector<float>(V, 1.0f), 0.75f, (float)V, 1024, 16, 16);
        check("all 0.1", std::vector<float>(V, 0.1f), inf, 0.0f, 1024, 16, 16);
        check("all denormal", std::vector<float>(V, 1e-44f), inf, 0.0f, 1024, 16, 16);
        check("all zero", std::vector<float>(V, 0.0f), 0.5f, 0.0f, 1024, 16, 16);
        for (int rep = 0; rep < 30; rep++) {
            std::vector<float> v(V);
            const int span = 10 + (int)(rng() % 140);
            for (auto& x : v) {
                const int ex = -(int)(rng() % span);
                const unsigned bitsn = 1 + (unsigned)(rng() % 24);
                x = std::ldexp((float)((rng() & ((1u << bitsn) - 1)) | 1u), ex - (int)bitsn);
                if (rng() % 13 == 0) x = 0.0f;
                if (rng() % 17 == 0) x = std::ldexp(1.0f, -126 - (int)(rng() % 23));
            }
            check("wide", v, inf, 0.0f, 1024, 16, 16);
            check("wide odd block", v, inf, 0.0f, 96, 3, 3);
            const float sum = plain(v, inf, 0.0f).s;
            check("wide pick", v, (float)U01(rng), sum, 1024, 16, 16);
        }
    }
    printf("failures %d, most iterations of one pass %ld (kernel shape 1024 x 16: %ld)\n", failures, worst_iterations, worst_real);
    return failures ? 1 : 0;
}
