/*
 * Panama FFM (java.lang.foreign, JDK 22) bindings of libjlamahip.so -- the subset of include/jlama_hip.h the provider
 * uses, in the shape jextract emits (see jextract_jlama_hip.sh; the reference's twin is
 * jlama-native/src/main/java22/com/github/tjake/jlama/tensor/operations/cnative/NativeSimd.java:117-189).
 * Every C parameter is int / long / float / pointer, so every descriptor below is JAVA_INT / JAVA_LONG / JAVA_FLOAT /
 * ADDRESS; there are no structs by value and no upcalls.
 */
package com.github.tjake.jlama.tensor.operations.cnative;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_FLOAT;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemoryLayout;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

public final class NativeHip {
    private NativeHip() {}

    public static final int JH_OK = 0;
    public static final int JH_ERR_NO_DEVICE = -1;
    public static final int JH_ERR_OOM = -2;
    public static final int JH_ERR_UNSUPPORTED = -3;
    public static final int JH_ERR_INVALID = -4;
    public static final int JH_ERR_HIP = -5;
    public static final int JH_DT_F32 = 0, JH_DT_BF16 = 1, JH_DT_I8 = 2, JH_DT_Q4 = 3;

    private static final Linker LINKER = Linker.nativeLinker();
    // the library is loaded by HipTensorOperations' static initialiser (System.loadLibrary / JarSupport)
    private static final SymbolLookup LOOKUP = SymbolLookup.loaderLookup().or(LINKER.defaultLookup());

    private static MethodHandle h(String name, MemoryLayout res, MemoryLayout... args) {
        MemorySegment sym = LOOKUP.find(name).orElseThrow(() -> new UnsatisfiedLinkError("unresolved symbol: " + name));
        return LINKER.downcallHandle(sym, res == null ? FunctionDescriptor.ofVoid(args) : FunctionDescriptor.of(res, args));
    }

    private static MemoryLayout[] sig(String s) {
        // i = int, l = long, f = float, p = pointer
        MemoryLayout[] out = new MemoryLayout[s.length()];
        for (int k = 0; k < s.length(); k++) {
            switch (s.charAt(k)) {
                case 'i': out[k] = JAVA_INT; break;
                case 'l': out[k] = JAVA_LONG; break;
                case 'f': out[k] = JAVA_FLOAT; break;
                default: out[k] = ADDRESS;
            }
        }
        return out;
    }

    private static final MethodHandle jh_init = h("jh_init", JAVA_INT, sig("ip"));
    private static final MethodHandle jh_name = h("jh_name", ADDRESS);
    private static final MethodHandle jh_last_error = h("jh_last_error", ADDRESS);
    private static final MethodHandle jh_parallel_split_size = h("jh_parallel_split_size", JAVA_INT);
    private static final MethodHandle jh_preferred_working_qtype = h("jh_preferred_working_qtype", JAVA_INT);
    private static final MethodHandle jh_register_tensor = h("jh_register_tensor", JAVA_LONG, sig("pl"));
    // int jh_gemm_q8_q4(long b_id, long bf_id, af, a, int aoffset, bf, b, int boffset, r, int roffset, m, n0, n, k, lda, ldaf, ldb, ldbf, ldc)
    private static final MethodHandle jh_gemm_q8_q4 = h("jh_gemm_q8_q4", JAVA_INT, sig("llppippipiiiiiiiiii"));
    private static final MethodHandle jh_gemm_f32_q4 = h("jh_gemm_f32_q4", JAVA_INT, sig("llpippipiiiiiiiii"));
    private static final MethodHandle jh_gemm_f32 = h("jh_gemm_f32", JAVA_INT, sig("lpipipiiiiiiii"));
    private static final MethodHandle jh_gemm_bf16 = h("jh_gemm_bf16", JAVA_INT, sig("lpipippiiiiiiii"));
    private static final MethodHandle jh_gemm_f32_bf16 = h("jh_gemm_f32_bf16", JAVA_INT, sig("lpipippiiiiiiii"));
    private static final MethodHandle jh_gemm_q8_q4_batch = h("jh_gemm_q8_q4_batch", JAVA_INT, sig("ippppippipiiiiiiiiii"));
    private static final MethodHandle jh_gemm_f32_q4_batch = h("jh_gemm_f32_q4_batch", JAVA_INT, sig("ipppippipiiiiiiiii"));
    private static final MethodHandle jh_gemm_f32_batch = h("jh_gemm_f32_batch", JAVA_INT, sig("ippipipiiiiiiii"));
    private static final MethodHandle jh_gemm_bf16_batch = h("jh_gemm_bf16_batch", JAVA_INT, sig("ippipippiiiiiiii"));
    private static final MethodHandle jh_gemm_f32_bf16_batch = h("jh_gemm_f32_bf16_batch", JAVA_INT, sig("ippipippiiiiiiii"));
    private static final MethodHandle jh_accumulate_f32 = h("jh_accumulate_f32", JAVA_INT, sig("ppii"));
    private static final MethodHandle jh_accumulate_f32_q4 = h("jh_accumulate_f32_q4", JAVA_INT, sig("pppii"));
    private static final MethodHandle jh_maccumulate_f32 = h("jh_maccumulate_f32", JAVA_INT, sig("ppii"));
    private static final MethodHandle jh_scale_f32 = h("jh_scale_f32", JAVA_INT, sig("fpii"));
    private static final MethodHandle jh_saxpy_f32 = h("jh_saxpy_f32", JAVA_INT, sig("fppiii"));
    private static final MethodHandle jh_saxpy_batch_f32 = h("jh_saxpy_batch_f32", JAVA_INT, sig("ppipiiiiii"));
    private static final MethodHandle jh_quantize_q8 = h("jh_quantize_q8", JAVA_INT, sig("piiiipipi"));
    private static final MethodHandle jh_quantize_bf16 = h("jh_quantize_bf16", JAVA_INT, sig("plp"));

    private static RuntimeException rethrow(Throwable t) {
        if (t instanceof RuntimeException) return (RuntimeException) t;
        if (t instanceof Error) throw (Error) t;
        return new AssertionError("should not reach here", t);
    }

    private static String cstr(MemorySegment p) {
        return p.equals(MemorySegment.NULL) ? "" : p.reinterpret(4096).getString(0);
    }

    public static int jh_init(int device, MemorySegment outInfo) {
        try { return (int) jh_init.invokeExact(device, outInfo); } catch (Throwable t) { throw rethrow(t); }
    }

    public static String jh_name() {
        try { return cstr((MemorySegment) jh_name.invokeExact()); } catch (Throwable t) { throw rethrow(t); }
    }

    public static String jh_last_error() {
        try { return cstr((MemorySegment) jh_last_error.invokeExact()); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_parallel_split_size() {
        try { return (int) jh_parallel_split_size.invokeExact(); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_preferred_working_qtype() {
        try { return (int) jh_preferred_working_qtype.invokeExact(); } catch (Throwable t) { throw rethrow(t); }
    }

    public static long jh_register_tensor(MemorySegment host, long bytes) {
        try { return (long) jh_register_tensor.invokeExact(host, bytes); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_q8_q4(long bId, long bfId, MemorySegment af, MemorySegment a, int aoffset, MemorySegment bf, MemorySegment b,
        int boffset, MemorySegment r, int roffset, int m, int n0, int n, int k, int lda, int ldaf, int ldb, int ldbf, int ldc) {
        try {
            return (int) jh_gemm_q8_q4.invokeExact(bId, bfId, af, a, aoffset, bf, b, boffset, r, roffset, m, n0, n, k, lda, ldaf, ldb, ldbf, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_f32_q4(long bId, long bfId, MemorySegment a, int aoffset, MemorySegment bf, MemorySegment b, int boffset,
        MemorySegment r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldbf, int ldc) {
        try {
            return (int) jh_gemm_f32_q4.invokeExact(bId, bfId, a, aoffset, bf, b, boffset, r, roffset, m, n0, n, k, lda, ldb, ldbf, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_f32(long bId, MemorySegment a, int aoffset, MemorySegment b, int boffset, MemorySegment r, int roffset, int m,
        int n0, int n, int k, int lda, int ldb, int ldc) {
        try {
            return (int) jh_gemm_f32.invokeExact(bId, a, aoffset, b, boffset, r, roffset, m, n0, n, k, lda, ldb, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_bf16(long bId, MemorySegment a, int aoffset, MemorySegment b, int boffset, MemorySegment cr, MemorySegment r,
        int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
        try {
            return (int) jh_gemm_bf16.invokeExact(bId, a, aoffset, b, boffset, cr, r, roffset, m, n0, n, k, lda, ldb, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_f32_bf16(long bId, MemorySegment a, int aoffset, MemorySegment b, int boffset, MemorySegment cr, MemorySegment r,
        int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
        try {
            return (int) jh_gemm_f32_bf16.invokeExact(bId, a, aoffset, b, boffset, cr, r, roffset, m, n0, n, k, lda, ldb, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_q8_q4_batch(int batchNum, MemorySegment bIds, MemorySegment bfIds, MemorySegment af, MemorySegment a, int aoffset,
        MemorySegment bf, MemorySegment b, int boffset, MemorySegment r, int roffset, int m, int n0, int n, int k, int lda, int ldaf, int ldb,
        int ldbf, int ldc) {
        try {
            return (int) jh_gemm_q8_q4_batch.invokeExact(batchNum, bIds, bfIds, af, a, aoffset, bf, b, boffset, r, roffset, m, n0, n, k, lda, ldaf,
                ldb, ldbf, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_f32_q4_batch(int batchNum, MemorySegment bIds, MemorySegment bfIds, MemorySegment a, int aoffset, MemorySegment bf,
        MemorySegment b, int boffset, MemorySegment r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldbf, int ldc) {
        try {
            return (int) jh_gemm_f32_q4_batch.invokeExact(batchNum, bIds, bfIds, a, aoffset, bf, b, boffset, r, roffset, m, n0, n, k, lda, ldb,
                ldbf, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_f32_batch(int batchNum, MemorySegment bIds, MemorySegment a, int aoffset, MemorySegment b, int boffset,
        MemorySegment r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
        try {
            return (int) jh_gemm_f32_batch.invokeExact(batchNum, bIds, a, aoffset, b, boffset, r, roffset, m, n0, n, k, lda, ldb, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_bf16_batch(int batchNum, MemorySegment bIds, MemorySegment a, int aoffset, MemorySegment b, int boffset,
        MemorySegment cr, MemorySegment r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
        try {
            return (int) jh_gemm_bf16_batch.invokeExact(batchNum, bIds, a, aoffset, b, boffset, cr, r, roffset, m, n0, n, k, lda, ldb, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_gemm_f32_bf16_batch(int batchNum, MemorySegment bIds, MemorySegment a, int aoffset, MemorySegment b, int boffset,
        MemorySegment cr, MemorySegment r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
        try {
            return (int) jh_gemm_f32_bf16_batch.invokeExact(batchNum, bIds, a, aoffset, b, boffset, cr, r, roffset, m, n0, n, k, lda, ldb, ldc);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_accumulate_f32(MemorySegment a, MemorySegment b, int offset, int length) {
        try { return (int) jh_accumulate_f32.invokeExact(a, b, offset, length); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_accumulate_f32_q4(MemorySegment a, MemorySegment nibRow, MemorySegment scaleRow, int offset, int length) {
        try { return (int) jh_accumulate_f32_q4.invokeExact(a, nibRow, scaleRow, offset, length); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_maccumulate_f32(MemorySegment a, MemorySegment b, int offset, int length) {
        try { return (int) jh_maccumulate_f32.invokeExact(a, b, offset, length); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_scale_f32(float factor, MemorySegment a, int offset, int length) {
        try { return (int) jh_scale_f32.invokeExact(factor, a, offset, length); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_saxpy_f32(float alpha, MemorySegment x, MemorySegment y, int xoffset, int yoffset, int limit) {
        try { return (int) jh_saxpy_f32.invokeExact(alpha, x, y, xoffset, yoffset, limit); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_saxpy_batch_f32(MemorySegment alpha, MemorySegment x, int ldx, MemorySegment y, int xoffset, int yoffset, int limit,
        int aoffset, int xrowoffset, int batchSize) {
        try {
            return (int) jh_saxpy_batch_f32.invokeExact(alpha, x, ldx, y, xoffset, yoffset, limit, aoffset, xrowoffset, batchSize);
        } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_quantize_q8(MemorySegment x, int rows, int ldx, int offset, int length, MemorySegment q, int ldq, MemorySegment d,
        int ldd) {
        try { return (int) jh_quantize_q8.invokeExact(x, rows, ldx, offset, length, q, ldq, d, ldd); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_quantize_bf16(MemorySegment x, long n, MemorySegment out) {
        try { return (int) jh_quantize_bf16.invokeExact(x, n, out); } catch (Throwable t) { throw rethrow(t); }
    }
}
