// A float running sum  s = fl(s + x_i), i ascending, x_i >= 0  evaluated EXACTLY by many lanes.
//
// AbstractModel.sample (AbstractModel.java:471-489) accumulates V = 128256 floats twice in index order (the sum of the
// exponentials, then the inverse-CDF walk); a lone lane needs ~4.5 cycles per dependent add, 0.24 ms per pass.  The order
// cannot change, but while s stays inside one binade the rounding of every step depends on s only through the PARITY of its
// significand:
//   s = M * U, U = 2^(k-23), k = max(exponent(s), -126), M < 2^24   (denormals and the first normal binade share U = 2^-149)
//   fl(s + x) = (M + q + r) * U  with  x / U = q + f,  r = [f > 1/2] + [f == 1/2 and M + q odd]      (round to nearest even)
// provided the result stays below 2^24 * U.  So a run of elements is a map  parity(M) -> increment of M  (SeqStep), and the
// hardware evaluates it by adding the elements to two REFERENCE values of the binade, 2^k (even significand) and 2^k + U (odd):
// the differences of the bit patterns are the increments (SeqRef).  Maps compose associatively (seq_compose): a prefix scan
// over the lanes' maps gives the partial sum behind every lane at once.  The first lane whose partial sum leaves the binade
// (or reaches the inverse-CDF threshold) is found from the scan; from the exact state in front of it that lane walks its own
// elements with plain float adds, and the scan resumes behind it in the new binade.
// tests/native/seqsum_harness.hip runs these functions on the host against the plain loop (wide exponent ranges, ties, denormals).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jh {

struct SeqStep { int d0, d1; };          // increment of M when the incoming M is even / odd
constexpr int SEQ_CAP = 1 << 26;         // saturation: anything >= 2^24 means "left the binade", exactness ends there
constexpr int SEQ_LIMIT = 1 << 24;

__host__ __device__ inline unsigned seq_bits(float x) { union { float f; unsigned u; } c; c.f = x; return c.u; }
__host__ __device__ inline float seq_float(unsigned b) { union { float f; unsigned u; } c; c.u = b; return c.f; }
__host__ __device__ inline int seq_min(int a, int b) { return a < b ? a : b; }

// s >= 0 finite: k and M of the header comment
__host__ __device__ inline int seq_k(float s) {
    const int e = (int)((seq_bits(s) >> 23) & 0xff);
    return e == 0 ? -126 : e - 127;
}
__host__ __device__ inline int seq_m(float s) {
    const unsigned b = seq_bits(s);
    return (int)(((b >> 23) & 0xff) == 0 ? (b & 0x7fffffu) : ((b & 0x7fffffu) | 0x800000u));
}
__host__ __device__ inline float seq_value(int m, int k) {   // m < 2^24; m < 2^23 only for k == -126
    return seq_float(m < 0x800000 ? (unsigned)m : (((unsigned)(k + 127) << 23) | ((unsigned)m & 0x7fffffu)));
}
// two reference runs through binade k: r0 starts at 2^k (even significand), r1 one unit above (odd)
struct SeqRef { float r0, r1; };
__host__ __device__ inline SeqRef seq_ref_begin(int k) {
    const unsigned b0 = (unsigned)(k + 127) << 23;
    SeqRef r;
    r.r0 = seq_float(b0);
    r.r1 = seq_float(b0 | 1u);
    return r;
}
__host__ __device__ inline void seq_ref_add(SeqRef& r, float x) { r.r0 += x; r.r1 += x; }
// the map of the elements added since seq_ref_begin.  A reference that reached 2^(k+1) has changed its unit: whatever M is, the
// true sum has left the binade as well (it is at least as large, except below 2^-126 where the cap says so explicitly).
__host__ __device__ inline SeqStep seq_ref_end(SeqRef r, int k) {
    const unsigned b0 = (unsigned)(k + 127) << 23, top = b0 + 0x800000u;
    SeqStep d;
    d.d0 = (int)(seq_bits(r.r0) - b0);
    d.d1 = (int)(seq_bits(r.r1) - (b0 | 1u));
    if (seq_bits(r.r1) >= top) d.d0 = d.d1 = SEQ_CAP;
    return d;
}
// first a, then b
__host__ __device__ inline SeqStep seq_compose(SeqStep a, SeqStep b) {
    SeqStep r;
    r.d0 = seq_min(SEQ_CAP, a.d0 + ((a.d0 & 1) ? b.d1 : b.d0));
    r.d1 = seq_min(SEQ_CAP, a.d1 + ((a.d1 & 1) ? b.d0 : b.d1));
    return r;
}
__host__ __device__ inline int seq_apply(int m, SeqStep a) { return seq_min(SEQ_CAP, m + ((m & 1) ? a.d1 : a.d0)); }
// smallest M with M * U >= u (the inverse-CDF stop, AbstractModel.java:484-487), clamped to [0, 2^24]; u = +inf -> 2^24
__host__ __device__ inline int seq_threshold(float u, int k) {
    if (!(u > 0.0f)) return 0;
    const double t = (double)u * __builtin_ldexp(1.0, 23 - k);       // exact: a float times a power of two in double range
    if (!(t < (double)SEQ_LIMIT)) return SEQ_LIMIT;
    const int c = (int)t;
    return (double)c < t ? c + 1 : c;
}

}  // namespace jh
