/*
 * JNI binding of libtsxform's C ABI (include/tsxform.h).  One native method per entry point the hot path needs;
 * java/jni/tsx_jni.c is the shim.  NOT compiled in this repository's CI (no JDK in the build image) - see INTEGRATION.md.
 */
package io.aiven.kafka.tieredstorage.gpu;

import java.nio.ByteBuffer;

public final class TsxNative {
    public static final int COMPRESS = 0x1;
    public static final int ENCRYPT = 0x2;
    public static final int CRC = 0x4;

    public static final int OK = 0;
    public static final int E_TAG_MISMATCH = -5;
    public static final int E_BAD_FRAME = -6;
    public static final int E_BAD_SIZE = -7;

    /** Size of one tsx_chunk_desc (include/tsxform.h), written/read through a direct little-endian ByteBuffer. */
    public static final int DESC_BYTES = 48;
    public static final int DESC_SRC_OFF = 0;
    public static final int DESC_DST_OFF = 8;
    public static final int DESC_SRC_LEN = 16;
    public static final int DESC_DST_CAP = 20;
    public static final int DESC_DST_LEN = 24;
    public static final int DESC_CRC32C = 28;
    public static final int DESC_STATUS = 32;
    public static final int DESC_IV = 36;

    /**
     * TSX_ZSTD_PROFILE_*: which libzstd behaviour the compressor reproduces (INTEGRATION.md, "Zstd profile").  1_5_7 is byte-identical to
     * the real 1.5.7; 1_5_6 is 1.5.7 without its pre-block splitter - an unverified stand-in for the 1.5.6 inside zstd-jni 1.5.6-9.
     */
    public static final int ZSTD_PROFILE_1_5_6 = 0;
    public static final int ZSTD_PROFILE_1_5_7 = 1;

    static {
        System.loadLibrary("tsxform_jni");   // links libtsxform.so
        final int devices = init();
        if (devices <= 0) {
            // there is no CPU implementation behind this binding
            throw new UnsatisfiedLinkError("tsxform: no usable gfx950 device: " + strerror(devices));
        }
    }

    private TsxNative() {
    }

    /** tsx_init(0, NULL): all visible devices. */
    private static native int init();

    public static native String strerror(int code);

    /** tsx_device_count: GPUs the library drives (all visible ones). */
    public static native int deviceCount();

    /** tsx_set_thread_device: device of the calling thread's batches, -1 = the least loaded one. */
    public static native int setThreadDevice(int device);

    /** tsx_host_register / tsx_host_unregister: pin a direct buffer that is reused for batches. */
    public static native int hostRegister(ByteBuffer buf);

    public static native int hostUnregister(ByteBuffer buf);

    /**
     * Per-thread, grow-only direct buffers (src, dst, descriptors) of the calling thread, pinned once per (re)allocation:
     * no off-heap allocation and no page faults on the hot path, and the copies go by DMA.
     */
    public static final class Buffers {
        private static final ThreadLocal<Buffers> LOCAL = ThreadLocal.withInitial(Buffers::new);
        private ByteBuffer src;
        private ByteBuffer dst;
        private ByteBuffer descs;

        public static Buffers get() {
            return LOCAL.get();
        }

        private static ByteBuffer grow(final ByteBuffer old, final long need, final boolean pin) {
            if (need > Integer.MAX_VALUE - 64) {
                // one direct ByteBuffer holds < 2 GiB: the batch size is bounded at construction time of the enumerations
                throw new IllegalArgumentException("batch of " + need + " bytes exceeds a direct ByteBuffer");
            }
            if (old != null && old.capacity() >= need) {
                old.clear();
                return old;
            }
            if (old != null && pin) {
                hostUnregister(old);
            }
            final ByteBuffer fresh = ByteBuffer.allocateDirect((int) Math.min(need + need / 8 + 64, (long) Integer.MAX_VALUE - 64))
                .order(java.nio.ByteOrder.LITTLE_ENDIAN);          // headroom for the next, slightly larger batch - never beyond what one buffer holds
            if (pin) {
                hostRegister(fresh);          // best effort: an unpinned buffer still works (runtime-staged copies)
            }
            return fresh;
        }

        public ByteBuffer src(final long need) {
            src = grow(src, need, true);
            return src;
        }

        public ByteBuffer dst(final long need) {
            dst = grow(dst, need, true);
            return dst;
        }

        public ByteBuffer descs(final int n) {
            descs = grow(descs, (long) n * DESC_BYTES, false);
            return descs;
        }

        /**
         * For a thread that is about to end (the broker's upload / fetch threads and the read-ahead helpers never do): the
         * pinning must be undone before the garbage collector frees the buffers' memory.
         */
        public static void release() {
            final Buffers b = LOCAL.get();
            if (b.src != null) {
                hostUnregister(b.src);
            }
            if (b.dst != null) {
                hostUnregister(b.dst);
            }
            b.src = null;
            b.dst = null;
            b.descs = null;
            LOCAL.remove();
        }
    }

    /** tsx_transformed_bound. */
    public static native long transformedBound(long n, int flags);

    /**
     * tsx_transform_batch / tsx_detransform_batch over direct buffers (host memory, TSX_MEM_HOST) with a pooled context.
     *
     * @param descs n * DESC_BYTES, little-endian, in/out
     * @param key   32-byte AES key or null; zeroised in native memory before returning
     * @return batch-level status (0 or a negative TSX_E_* code); per-chunk status is in descs
     */
    public static native int transformBatch(int flags, byte[] key, byte[] aad, int zstdProfile,
                                            ByteBuffer descs, int n, ByteBuffer src, ByteBuffer dst);

    /**
     * tsx_transform_batch with TSX_MEM_HOST_PACKED: the transformed chunks are written back to back into {@code dst} - the
     * upload's own buffer (a multipart part, TransformFinisher's object) - and each descriptor returns its chunk's offset and
     * size; a chunk that does not fit any more comes back with TSX_E_DST_TOO_SMALL.
     */
    public static native int transformBatchPacked(int flags, byte[] key, byte[] aad, int zstdProfile,
                                                  ByteBuffer descs, int n, ByteBuffer src, ByteBuffer dst);

    public static native int detransformBatch(int flags, byte[] key, byte[] aad,
                                              ByteBuffer descs, int n, ByteBuffer src, ByteBuffer dst);
}
