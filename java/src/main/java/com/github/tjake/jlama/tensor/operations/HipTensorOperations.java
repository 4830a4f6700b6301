/*
 * HipTensorOperations -- the MI355X (gfx950) TensorOperations provider of jlama-hip.
 *
 * Drop-in sibling of NativeSimdTensorOperations / NativeGPUTensorOperations
 * (jlama-native/src/main/java/com/github/tjake/jlama/tensor/operations/): same interface
 * (jlama-core/.../tensor/operations/TensorOperations.java:25-161), same argument marshaling as
 * NativeSimdTensorOperations.java:84-436 -- libjlamahip.so keeps the offset / stride conventions of
 * jlama-native/src/main/c/simd/vector_simd.h:22-38 on purpose -- plus weight registration in HBM like
 * NativeGPUTensorOperations.registerModelTensor (:104-151).
 *
 * Place under jlama-native/src/main/java/ (bindings: cnative/NativeHip.java under src/main/java22/), ship
 * libjlamahip.so in META-INF/native/lib/ and apply TensorOperationsProvider.patch.  This file cannot be compiled in the
 * jlama-hip build image (no JDK there); tests/test_java_binding.py checks the FFM descriptors against the C header.
 */
package com.github.tjake.jlama.tensor.operations;

import com.github.tjake.jlama.safetensors.DType;
import com.github.tjake.jlama.tensor.AbstractTensor;
import com.github.tjake.jlama.tensor.BFloat16BufferTensor;
import com.github.tjake.jlama.tensor.Q4ByteBufferTensor;
import com.github.tjake.jlama.tensor.Q8ByteBufferTensor;
import com.github.tjake.jlama.tensor.operations.cnative.NativeHip;
import com.github.tjake.jlama.tensor.operations.util.JarSupport;
import com.github.tjake.jlama.tensor.operations.util.MemorySegmentSupport;
import com.github.tjake.jlama.util.MachineSpec;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.ValueLayout;
import java.util.concurrent.ConcurrentHashMap;
import java.util.concurrent.ConcurrentMap;
import java.util.concurrent.atomic.AtomicBoolean;
import java.util.concurrent.atomic.AtomicLong;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;

public class HipTensorOperations implements TensorOperations {
    private static final Logger logger = LoggerFactory.getLogger(HipTensorOperations.class);

    static {
        // same discovery as the reference's native providers (JarSupport.java:34-71)
        if (!JarSupport.maybeLoadLibrary("jlamahip")) System.loadLibrary("jlamahip");
    }

    /** Element-wise ops on host tensors: Panama by default (they are PCIe round trips on a device); opt in to the device. */
    private static final boolean deviceElementwise = Boolean.getBoolean("jlama.hip_elementwise");

    private static final TensorOperations delegate;

    static {
        TensorOperations tmp;
        try {
            tmp = new PanamaTensorOperations(MachineSpec.VECTOR_TYPE);
        } catch (Throwable t) {
            tmp = new NaiveTensorOperations();
        }
        delegate = tmp;
    }

    /** uid -> id of the HBM copy (jh_register_tensor); Q4 tensors register nibbles and blockF separately. */
    private final ConcurrentMap<String, Long> tensorCache = new ConcurrentHashMap<>();
    private final AtomicBoolean limitReached = new AtomicBoolean(false);
    private final AtomicLong totalBytesAllocated = new AtomicLong(0);

    /** per-thread arrays of weight / scale ids for the `_batch` calls (dotProductBatchChunk is called from pfor workers) */
    private static final int MAX_BATCH = 16;
    private final ThreadLocal<MemorySegment[]> idArrays = ThreadLocal.withInitial(() -> new MemorySegment[] {
        Arena.global().allocate(ValueLayout.JAVA_LONG, MAX_BATCH), Arena.global().allocate(ValueLayout.JAVA_LONG, MAX_BATCH) });

    public HipTensorOperations() {
        int device = Integer.getInteger("jlama.hip_device", 0);
        // Throws when no MI355X is usable so that TensorOperationsProvider falls through (TensorOperationsProvider.java:50-87)
        check(NativeHip.jh_init(device, MemorySegment.NULL), "jh_init");
        logger.info("{} initialised on device {}", NativeHip.jh_name(), device);
    }

    private static void check(int rc, String what) {
        if (rc == NativeHip.JH_OK) return;
        String msg = what + ": " + NativeHip.jh_last_error() + " (" + rc + ")";
        if (rc == NativeHip.JH_ERR_UNSUPPORTED) throw new UnsupportedOperationException(msg);   // PanamaTensorOperations.java:125-142
        if (rc == NativeHip.JH_ERR_INVALID) throw new IllegalArgumentException(msg);
        throw new RuntimeException(msg);
    }

    @Override
    public String name() {
        return NativeHip.jh_name();
    }

    /** Every GEMM arrives whole: the device parallelises internally (NativeGPUTensorOperations.java:98-101). */
    @Override
    public int parallelSplitSize() {
        return NativeHip.jh_parallel_split_size();
    }

    @Override
    public DType preferredWorkingQuantizedType() {
        return NativeHip.jh_preferred_working_qtype() == NativeHip.JH_DT_I8 ? DType.I8 : DType.F32;
    }

    // ------------------------------------------------------------------------------------------ weight registration
    @Override
    public void registerModelTensor(AbstractTensor t) {
        if (tensorCache.containsKey(t.getUid()) || limitReached.get()) return;
        long byteSize = t.getMemorySegment().byteSize();
        try {
            tensorCache.computeIfAbsent(t.getUid(), s -> registerSegment(t.getMemorySegment()));
            if (t.dType() == DType.Q4) {
                Q4ByteBufferTensor q4 = (Q4ByteBufferTensor) t;
                byteSize += q4.getBlockF().getMemorySegment().byteSize();
                tensorCache.computeIfAbsent(q4.getBlockF().getUid(), s -> registerSegment(q4.getBlockF().getMemorySegment()));
            }
            totalBytesAllocated.addAndGet(byteSize);
        } catch (RuntimeException r) {
            // JH_ERR_OOM: keep the tensor on the host, later calls ship it per call (NativeGPUTensorOperations.java:141-149)
            tensorCache.remove(t.getUid());
            limitReached.set(true);
            logger.warn("HBM limit reached after {} bytes, remaining tensors stay on the host", totalBytesAllocated.get());
        }
    }

    private Long registerSegment(MemorySegment seg) {
        synchronized (tensorCache) {
            long id = NativeHip.jh_register_tensor(seg, seg.byteSize());
            if (id < 0) throw new RuntimeException("jh_register_tensor: " + NativeHip.jh_last_error());
            return id;
        }
    }

    private long idOf(AbstractTensor t) {
        Long id = tensorCache.get(t.getUid());
        return id == null ? -1L : id;
    }

    // ------------------------------------------------------------------------------------------ batchDotProduct
    @Override
    public void batchDotProduct(
        AbstractTensor result,
        AbstractTensor at,
        AbstractTensor bt,
        int aColumnOffset,
        int bColumnOffset,
        int columnLength,
        int rRowOffset,
        int bRowOffset,
        int rowChunkSize
    ) {
        int M = at.shape().dim(0);
        int N = rowChunkSize;
        int K = columnLength;

        // identical to NativeSimdTensorOperations.java:100-107
        int aOffset = at.getOffset(0, aColumnOffset);
        int bOffset = bt.getOffset(bt.shape().sparseRowOffset(), bColumnOffset);
        int rOffset = result.shape().sparseColumnOffset() - bt.shape().sparseRowOffset() - rRowOffset;
        int adjBRowOffset = bRowOffset - bt.shape().sparseRowOffset();

        MemorySegment cr = result.dType() == DType.BF16 ? result.getMemorySegment() : MemorySegment.NULL;
        MemorySegment rf = result.dType() == DType.F32 ? result.getMemorySegment() : MemorySegment.NULL;
        int rc;
        switch (at.dType()) {
            case BF16:
                switch (bt.dType()) {
                    case BF16:
                        rc = NativeHip.jh_gemm_bf16(idOf(bt), at.getMemorySegment(), aOffset, bt.getMemorySegment(), bOffset, cr, rf, rOffset, M,
                            adjBRowOffset, N, K, at.getStride(), bt.getStride(), result.getStride());
                        break;
                    default:
                        throw new UnsupportedOperationException(at.dType().name() + " " + bt.dType().name());
                }
                break;
            case F32:
                switch (bt.dType()) {
                    case F32:
                        rc = NativeHip.jh_gemm_f32(idOf(bt), at.getMemorySegment(), aOffset, bt.getMemorySegment(), bOffset,
                            result.getMemorySegment(), rOffset, M, adjBRowOffset, N, K, at.getStride(), bt.getStride(), result.getStride());
                        break;
                    case BF16:
                        rc = NativeHip.jh_gemm_f32_bf16(idOf(bt), at.getMemorySegment(), aOffset, bt.getMemorySegment(), bOffset, cr, rf, rOffset,
                            M, adjBRowOffset, N, K, at.getStride(), bt.getStride(), result.getStride());
                        break;
                    case Q4: {
                        Q4ByteBufferTensor b = (Q4ByteBufferTensor) bt;
                        rc = NativeHip.jh_gemm_f32_q4(idOf(b), idOf(b.getBlockF()), at.getMemorySegment(), aOffset,
                            b.getBlockF().getMemorySegment(), b.getMemorySegment(), b.getMemorySegmentOffset(bOffset), result.getMemorySegment(),
                            rOffset, M, adjBRowOffset, N, K, at.getStride(), b.getMemorySegmentOffset(b.getStride()), b.getBlockF().getStride(),
                            result.getStride());
                        break;
                    }
                    default:
                        throw new UnsupportedOperationException(at.dType().name() + " " + bt.dType().name());
                }
                break;
            case I8:
                switch (bt.dType()) {
                    case Q4: {
                        Q8ByteBufferTensor a = (Q8ByteBufferTensor) at;
                        Q4ByteBufferTensor b = (Q4ByteBufferTensor) bt;
                        rc = NativeHip.jh_gemm_q8_q4(idOf(b), idOf(b.getBlockF()), a.getBlockF().getMemorySegment(), a.getMemorySegment(), aOffset,
                            b.getBlockF().getMemorySegment(), b.getMemorySegment(), b.getMemorySegmentOffset(bOffset), result.getMemorySegment(),
                            rOffset, M, adjBRowOffset, N, K, a.getStride(), a.getBlockF().getStride(), b.getMemorySegmentOffset(b.getStride()),
                            b.getBlockF().getStride(), result.getStride());
                        break;
                    }
                    default:
                        throw new UnsupportedOperationException(at.dType().name() + " " + bt.dType().name());
                }
                break;
            default:
                throw new UnsupportedOperationException(at.dType().name());
        }
        check(rc, "batchDotProduct");
    }

    // ------------------------------------------------------------------------------------------ dotProductBatchChunk
    @Override
    public void dotProductBatchChunk(
        AbstractTensor[] r,
        AbstractTensor a,
        AbstractTensor[] b,
        int columnOffset,
        int columnLength,
        int bRowOffset,
        int rowChunkSize
    ) {
        if (r.length > MAX_BATCH) {
            TensorOperations.super.dotProductBatchChunk(r, a, b, columnOffset, columnLength, bRowOffset, rowChunkSize);
            return;
        }
        // arrays of result / weight / scale pointers (NativeSimdTensorOperations.java:236-247)
        MemorySegment[] tmp = MemorySegmentSupport.setupBatch(
            i -> r[i].getMemorySegment(),
            i -> b[i].getMemorySegment(),
            i -> b[i] instanceof Q4ByteBufferTensor ? ((Q4ByteBufferTensor) b[i]).getBlockF().getMemorySegment() : MemorySegment.NULL,
            r.length
        );
        MemorySegment ra = tmp[0];
        MemorySegment rb = tmp[1];
        MemorySegment rc = tmp[2];
        MemorySegment[] ids = idArrays.get();
        for (int i = 0; i < r.length; i++) {
            ids[0].setAtIndex(ValueLayout.JAVA_LONG, i, idOf(b[i]));
            ids[1].setAtIndex(ValueLayout.JAVA_LONG, i, b[i] instanceof Q4ByteBufferTensor ? idOf(((Q4ByteBufferTensor) b[i]).getBlockF()) : -1L);
        }

        int M = a.shape().dim(0);
        int N = rowChunkSize;
        int K = columnLength;
        int aOffset = a.getOffset(0, columnOffset);
        int bOffset = b[0].getOffset(b[0].shape().sparseRowOffset(), columnOffset);
        int adjBRowOffset = bRowOffset - b[0].shape().sparseRowOffset();
        int rOffset = r[0].shape().sparseColumnOffset() - b[0].shape().sparseRowOffset();
        MemorySegment cr = r[0].dType() == DType.BF16 ? ra : MemorySegment.NULL;
        MemorySegment rf = r[0].dType() == DType.F32 ? ra : MemorySegment.NULL;

        int status;
        switch (a.dType()) {
            case BF16:
                switch (b[0].dType()) {
                    case BF16:
                        status = NativeHip.jh_gemm_bf16_batch(r.length, ids[0], a.getMemorySegment(), aOffset, rb, bOffset, cr, rf, rOffset, M,
                            adjBRowOffset, N, K, a.getStride(), b[0].getStride(), r[0].getStride());
                        break;
                    default:
                        throw new UnsupportedOperationException(a.dType().name() + " " + b[0].dType().name());
                }
                break;
            case F32:
                switch (b[0].dType()) {
                    case F32:
                        status = NativeHip.jh_gemm_f32_batch(r.length, ids[0], a.getMemorySegment(), aOffset, rb, bOffset, ra, rOffset, M,
                            adjBRowOffset, N, K, a.getStride(), b[0].getStride(), r[0].getStride());
                        break;
                    case BF16:
                        status = NativeHip.jh_gemm_f32_bf16_batch(r.length, ids[0], a.getMemorySegment(), aOffset, rb, bOffset, cr, rf, rOffset, M,
                            adjBRowOffset, N, K, a.getStride(), b[0].getStride(), r[0].getStride());
                        break;
                    case Q4: {
                        Q4ByteBufferTensor bt = (Q4ByteBufferTensor) b[0];
                        status = NativeHip.jh_gemm_f32_q4_batch(r.length, ids[0], ids[1], a.getMemorySegment(), aOffset, rc, rb,
                            bt.getMemorySegmentOffset(bOffset), ra, rOffset, M, adjBRowOffset, N, K, a.getStride(),
                            bt.getMemorySegmentOffset(bt.getStride()), bt.getBlockF().getStride(), r[0].getStride());
                        break;
                    }
                    default:
                        throw new UnsupportedOperationException(a.dType().name() + " " + b[0].dType().name());
                }
                break;
            case I8:
                switch (b[0].dType()) {
                    case Q4: {
                        Q8ByteBufferTensor at = (Q8ByteBufferTensor) a;
                        Q4ByteBufferTensor bt = (Q4ByteBufferTensor) b[0];
                        status = NativeHip.jh_gemm_q8_q4_batch(r.length, ids[0], ids[1], at.getBlockF().getMemorySegment(), a.getMemorySegment(),
                            aOffset, rc, rb, bt.getMemorySegmentOffset(bOffset), ra, rOffset, M, adjBRowOffset, N, K, a.getStride(),
                            at.getBlockF().getStride(), bt.getMemorySegmentOffset(bt.getStride()), bt.getBlockF().getStride(), r[0].getStride());
                        break;
                    }
                    default:
                        throw new UnsupportedOperationException(a.dType().name() + " " + b[0].dType().name());
                }
                break;
            default:
                throw new UnsupportedOperationException(a.dType().name());
        }
        check(status, "dotProductBatchChunk");
    }

    // ------------------------------------------------------------------------------------------ element-wise
    // Row `i` of a dense 2-D F32 tensor as a segment starting at the row (the C entry points take offsets inside the row).
    private static MemorySegment row(AbstractTensor t, int i) {
        return t.getMemorySegment().asSlice(t.getMemorySegmentOffset(t.getOffset(i, 0)));
    }

    @Override
    public void accumulate(AbstractTensor a, AbstractTensor b, int offset, int length) {
        boolean q4 = b.dType() == DType.Q4;
        if (!deviceElementwise || a.dType() != DType.F32 || !(b.dType() == DType.F32 || q4)) {
            delegate.accumulate(a, b, offset, length);
            return;
        }
        // per row of a; b is broadcast when it has one row (PanamaTensorOperations.java:2150-2218)
        boolean bBatch = b.shape().first() > 1;
        for (int ai = 0; ai < a.shape().first(); ai++) {
            int bi = bBatch ? ai : 0;
            if (q4) {
                Q4ByteBufferTensor qb = (Q4ByteBufferTensor) b;
                check(NativeHip.jh_accumulate_f32_q4(row(a, ai), row(qb, bi), row(qb.getBlockF(), bi), offset, length), "accumulate");
            } else {
                check(NativeHip.jh_accumulate_f32(row(a, ai), row(b, bi), offset, length), "accumulate");
            }
        }
    }

    @Override
    public void maccumulate(AbstractTensor a, AbstractTensor b, int offset, int length) {
        if (!deviceElementwise || a.dType() != DType.F32 || b.dType() != DType.F32) {
            delegate.maccumulate(a, b, offset, length);
            return;
        }
        boolean bBatch = b.shape().first() > 1;
        for (int ai = 0; ai < a.shape().first(); ai++)
            check(NativeHip.jh_maccumulate_f32(row(a, ai), row(b, bBatch ? ai : 0), offset, length), "maccumulate");
    }

    @Override
    public void saxpy(float alpha, AbstractTensor x, AbstractTensor y, int xoffset, int yoffset, int limit) {
        if (!deviceElementwise || x.dType() != DType.F32 || y.dType() != DType.F32) {
            delegate.saxpy(alpha, x, y, xoffset, yoffset, limit);
            return;
        }
        check(NativeHip.jh_saxpy_f32(alpha, row(x, 0), row(y, 0), xoffset, yoffset, limit), "saxpy");
    }

    @Override
    public void saxpy(
        AbstractTensor alpha,
        AbstractTensor x,
        AbstractTensor y,
        int xoffset,
        int yoffset,
        int limit,
        int aOffset,
        int xRowOffset,
        int batchSize
    ) {
        if (!deviceElementwise || x.dType() != DType.F32 || y.dType() != DType.F32 || alpha.dType() != DType.F32) {
            delegate.saxpy(alpha, x, y, xoffset, yoffset, limit, aOffset, xRowOffset, batchSize);
            return;
        }
        // y += sum_n alpha[aOffset+n] * x[xRowOffset+n, xoffset..] (TensorOperations.java:119-135): one call, one PCIe round trip
        check(NativeHip.jh_saxpy_batch_f32(row(alpha, 0), x.getMemorySegment(), x.getStride(), row(y, 0), xoffset, yoffset, limit, aOffset,
            xRowOffset, batchSize), "saxpy");
    }

    @Override
    public void scale(float factor, AbstractTensor x, int offset, int length) {
        if (!deviceElementwise || x.dType() != DType.F32) {
            delegate.scale(factor, x, offset, length);
            return;
        }
        for (int i = 0; i < x.shape().first(); i++) check(NativeHip.jh_scale_f32(factor, row(x, i), offset, length), "scale");
    }

    /** F32 -> I8 with the Panama-512 quantizer semantics (PanamaTensorOperations.java:1684-1723) or F32 -> BF16 (RNE). */
    @Override
    public AbstractTensor quantize(AbstractTensor t, DType qtype, int offset, int length) {
        if (!deviceElementwise || t.dType() != DType.F32 || t.shape().isSparse()) return delegate.quantize(t, qtype, offset, length);
        int rows = t.shape().first();
        int cols = t.shape().last();
        switch (qtype) {
            case I8: {
                if (offset % Q8ByteBufferTensor.BLOCK_SIZE != 0 || length % Q8ByteBufferTensor.BLOCK_SIZE != 0) break;
                Q8ByteBufferTensor q = new Q8ByteBufferTensor(t.shape());
                check(NativeHip.jh_quantize_q8(t.getMemorySegment(), rows, t.getStride(), offset, length, q.getMemorySegment(), cols,
                    q.getBlockF().getMemorySegment(), q.getBlockF().getStride()), "quantize");
                return q;
            }
            case BF16: {
                if (offset != 0 || length != cols) break;
                BFloat16BufferTensor h = new BFloat16BufferTensor(t.shape());
                check(NativeHip.jh_quantize_bf16(t.getMemorySegment(), (long) rows * cols, h.getMemorySegment()), "quantize");
                return h;
            }
            default:
                break;
        }
        return delegate.quantize(t, qtype, offset, length);
    }
}
