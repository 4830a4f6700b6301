/*
 * Tier 2 of jlama-hip on the Java side: a Llama / Mistral checkpoint resident on one MI355X (or a layer shard of it), driven by
 * the unchanged Jlama host.  Where LlamaModel (jlama-core/.../model/llama/LlamaModel.java:67-173) builds TransformerBlocks out
 * of host tensors and AbstractModel.generate() (AbstractModel.java:515-646) walks them token by token, this class hands the
 * same tensors to libjlamahip.so once (jh_model_set_weight: the Q4 nibble segment and its blockF segment, exactly the two
 * segments NativeGPUTensorOperations.registerModelTensor uploads, NativeGPUTensorOperations.java:104-151) and then forwards
 * whole calls: batchForward -> jh_forward, sample(T = 0) -> jh_sample, the greedy loop -> jh_decode_n (one hipGraph replay per
 * token, stop tokens honoured on the device).  Tokenizer, prompt templating, sampling with T > 0 and the response plumbing stay
 * in AbstractModel.  Not compilable in this repository's container (no JDK); the call sequence is the one
 * jlama_amd/model.py makes through the same C ABI, which is what the GPU tests exercise.
 */
package com.github.tjake.jlama.model.hip;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_FLOAT;
import static java.lang.foreign.ValueLayout.JAVA_INT;

import com.github.tjake.jlama.model.DistributedContext;
import com.github.tjake.jlama.safetensors.Config;
import com.github.tjake.jlama.safetensors.DType;
import com.github.tjake.jlama.safetensors.WeightLoader;
import com.github.tjake.jlama.tensor.AbstractTensor;
import com.github.tjake.jlama.tensor.Q4ByteBufferTensor;
import com.github.tjake.jlama.tensor.operations.cnative.NativeHip;
import com.github.tjake.jlama.tensor.operations.cnative.NativeHipModel;
import java.lang.foreign.Arena;
import java.lang.foreign.MemoryLayout;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.StructLayout;
import java.util.List;

public final class HipResidentLlama implements AutoCloseable {
    /** jh_config of include/jlama_hip.h, field for field. */
    static final StructLayout JH_CONFIG = MemoryLayout.structLayout(
        JAVA_INT.withName("embedding_length"), JAVA_INT.withName("hidden_length"), JAVA_INT.withName("n_heads"),
        JAVA_INT.withName("n_kv_heads"), JAVA_INT.withName("head_size"), JAVA_INT.withName("n_layers"), JAVA_INT.withName("vocab_size"),
        JAVA_INT.withName("context_length"), JAVA_INT.withName("weight_dtype"), JAVA_INT.withName("layer_start"),
        JAVA_INT.withName("layer_end"), JAVA_FLOAT.withName("rms_eps"), JAVA_FLOAT.withName("rope_theta"), JAVA_FLOAT.withName("rope_scaling"));

    static {   // the struct passed by pointer must be the struct the library was compiled with (size + every field offset)
        try (Arena a = Arena.ofConfined()) {
            MemorySegment lay = a.allocate(JAVA_INT, 32);
            int cnt = NativeHipModel.jh_abi_config_layout(lay, 32);
            List<MemoryLayout> fields = JH_CONFIG.memberLayouts();
            boolean ok = cnt == fields.size() + 1 && lay.getAtIndex(JAVA_INT, 0) == JH_CONFIG.byteSize();
            for (int i = 0; ok && i < fields.size(); i++)
                ok = lay.getAtIndex(JAVA_INT, i + 1) == JH_CONFIG.byteOffset(MemoryLayout.PathElement.groupElement(i));
            if (!ok) throw new IllegalStateException("jlama-hip: jh_config layout of libjlamahip.so differs from this binding");
        }
    }

    // weight slots and dtypes of include/jlama_hip.h
    static final int W_Q = 0, W_K = 1, W_V = 2, W_O = 3, W_GATE = 4, W_UP = 5, W_DOWN = 6, W_NORM1 = 7, W_NORM2 = 8, W_EMBED = 9,
        W_LMHEAD = 10, W_FINALNORM = 11;

    private final Arena arena = Arena.ofShared();
    private final Config c;
    private final MemorySegment model;

    public HipResidentLlama(Config c, WeightLoader weights, DType modelDType, double ropeTheta, double ropeScaling) {
        this.c = c;
        DistributedContext d = c.dctx();
        MemorySegment cfg = arena.allocate(JH_CONFIG);
        int[] ints = { c.embeddingLength, c.hiddenLength, c.numberOfHeads, c.numberOfKeyValueHeads, c.headSize, c.numberOfLayers,
            c.vocabularySize, c.contextLength, modelDType == DType.Q4 ? NativeHip.JH_DT_Q4 : NativeHip.JH_DT_BF16, d.layerStart, d.layerEnd };
        for (int i = 0; i < ints.length; i++) cfg.setAtIndex(JAVA_INT, i, ints[i]);
        cfg.set(JAVA_FLOAT, 44, c.layerNormEps);
        cfg.set(JAVA_FLOAT, 48, (float) ropeTheta);
        cfg.set(JAVA_FLOAT, 52, (float) ropeScaling);
        MemorySegment out = arena.allocate(ADDRESS);
        check(NativeHipModel.jh_model_create(cfg, out));
        this.model = out.get(ADDRESS, 0);
        // the tensors LlamaModel.loadTransformerBlockWeights / loadInputWeights / loadOutputWeights read, by the same names
        for (int i = d.layerStart; i < d.layerEnd; i++) {
            String base = "model.layers." + i + ".";
            set(i, W_Q, weights.load(base + "self_attn.q_proj.weight", d, true, false));
            set(i, W_K, weights.load(base + "self_attn.k_proj.weight", d, true, false));
            set(i, W_V, weights.load(base + "self_attn.v_proj.weight", d, true, false));
            set(i, W_O, weights.load(base + "self_attn.o_proj.weight", d, false, true));
            set(i, W_GATE, weights.load(base + "mlp.gate_proj.weight", d, true, false));
            set(i, W_UP, weights.load(base + "mlp.up_proj.weight", d, true, false));
            set(i, W_DOWN, weights.load(base + "mlp.down_proj.weight", d, false, true));
            set(i, W_NORM1, weights.load(base + "input_layernorm.weight"));
            set(i, W_NORM2, weights.load(base + "post_attention_layernorm.weight"));
        }
        if (d.layerStart == 0 || !weights.isWeightPresent("lm_head.weight")) set(-1, W_EMBED, weights.load("model.embed_tokens.weight"));
        if (d.layerEnd == c.numberOfLayers) {
            set(-1, W_FINALNORM, weights.load("model.norm.weight"));
            if (weights.isWeightPresent("lm_head.weight")) set(-1, W_LMHEAD, weights.load("lm_head.weight"));   // absent => tied (LlamaModel.java:155-158)
        }
    }

    /** One tensor: Q4 = nibble segment + blockF scales, BF16 / F32 = the tensor's own segment; the library copies it to HBM. */
    private void set(int layer, int slot, AbstractTensor t) {
        int rows = t.shape().dim(0), cols = t.shape().dims() > 1 ? t.shape().dim(1) : 1;
        if (t.shape().dims() == 1) { rows = 1; cols = t.shape().dim(0); }
        if (t instanceof Q4ByteBufferTensor q) {
            check(NativeHipModel.jh_model_set_weight(model, layer, slot, NativeHip.JH_DT_Q4, q.getMemorySegment(), q.getBlockF().getMemorySegment(), rows, cols, 0));
        } else {
            int dt = t.dType() == DType.BF16 ? NativeHip.JH_DT_BF16 : NativeHip.JH_DT_F32;
            check(NativeHipModel.jh_model_set_weight(model, layer, slot, dt, t.getMemorySegment(), MemorySegment.NULL, rows, cols, 0));
        }
    }

    /** KvBufferCache.getKvBuffer(session) (KvBufferCache.java:58-60): the KV pages of one conversation, in HBM. */
    public Session newSession(int maxContext) {
        MemorySegment out = arena.allocate(ADDRESS);
        check(NativeHipModel.jh_session_create(model, maxContext, 1L << 23, out));
        return new Session(out.get(ADDRESS, 0));
    }

    public final class Session implements AutoCloseable {
        private final MemorySegment s;

        Session(MemorySegment s) {
            this.s = s;
            List<Integer> eos = c.eosTokens;
            if (eos != null && !eos.isEmpty()) {
                MemorySegment ids = arena.allocate(JAVA_INT, eos.size());
                for (int i = 0; i < eos.size(); i++) ids.setAtIndex(JAVA_INT, i, eos.get(i));
                check(NativeHipModel.jh_session_set_eos(s, ids, eos.size()));      // AbstractModel.java:600-603
            }
        }

        /** AbstractModel.batchForward(tokens, startPos, kv) (AbstractModel.java:295-312); the rows stay on the device. */
        public void batchForward(int[] tokens, int startPos) {
            try (Arena a = Arena.ofConfined()) {
                MemorySegment t = a.allocateFrom(JAVA_INT, tokens);
                check(NativeHipModel.jh_forward(s, t, MemorySegment.NULL, tokens.length, startPos, MemorySegment.NULL));
            }
        }

        /** AbstractModel.sample(output, temperature, uniformSample, logits) (AbstractModel.java:443-491); the caller draws u. */
        public int sample(float temperature, float uniformSample) {
            try (Arena a = Arena.ofConfined()) {
                MemorySegment tok = a.allocate(JAVA_INT);
                check(NativeHipModel.jh_sample(s, temperature, uniformSample, tok, MemorySegment.NULL));
                return tok.get(JAVA_INT, 0);
            }
        }

        /** The greedy loop of AbstractModel.generate() (AbstractModel.java:590-621) when no per-token callback is needed. */
        public int[] decodeGreedy(int firstToken, int position, int maxTokens) {
            try (Arena a = Arena.ofConfined()) {
                MemorySegment out = a.allocate(JAVA_INT, maxTokens);
                check(NativeHipModel.jh_decode_n(s, firstToken, position, maxTokens, out));
                MemorySegment n = a.allocate(JAVA_INT);
                check(NativeHipModel.jh_decode_generated(s, n));                      // fewer than asked after a stop token
                return out.asSlice(0, 4L * n.get(JAVA_INT, 0)).toArray(JAVA_INT);
            }
        }

        /** The same loop at temperature > 0 (AbstractModel.java:471-489): softmax((l - max) / T), inverse CDF against
         *  uniforms[i] -- the caller draws them (the reference: ThreadLocalRandom.nextFloat() per token, :594). */
        public int[] decodeSampled(int firstToken, int position, int maxTokens, float temperature, float[] uniforms) {
            // the native side copies maxTokens floats and cannot see the array's length
            if (uniforms == null || uniforms.length < maxTokens)
                throw new IllegalArgumentException("decodeSampled: " + maxTokens + " tokens need as many uniforms, got " + (uniforms == null ? 0 : uniforms.length));
            try (Arena a = Arena.ofConfined()) {
                MemorySegment out = a.allocate(JAVA_INT, maxTokens);
                MemorySegment u = a.allocateFrom(JAVA_FLOAT, uniforms);
                check(NativeHipModel.jh_decode_n_sampled(s, firstToken, position, maxTokens, temperature, u, out));
                MemorySegment n = a.allocate(JAVA_INT);
                check(NativeHipModel.jh_decode_generated(s, n));
                return out.asSlice(0, 4L * n.get(JAVA_INT, 0)).toArray(JAVA_INT);
            }
        }

        /** Verification: the reference's Panama summation order, bit-identical ids and logits (DESIGN.md section 4). */
        public void setStrictOrder(boolean on) {
            check(NativeHipModel.jh_session_set_strict(s, on ? 1 : 0));
        }

        @Override
        public void close() {
            NativeHipModel.jh_session_destroy(s);
        }
    }

    private static void check(int rc) {
        if (rc < 0) throw new IllegalStateException("jlama-hip: " + NativeHip.jh_last_error() + " (" + rc + ")");
    }

    @Override
    public void close() {
        NativeHipModel.jh_model_destroy(model);
        arena.close();
    }
}
