/*
 * Replaces CompressionChunkEnumeration + EncryptionChunkEnumeration in RemoteStorageManager.transformation()
 * (core/.../RemoteStorageManager.java:434-453) with one batched call into libtsxform.  Same contract as the two it
 * replaces: one Zstd frame per chunk (CompressionChunkEnumeration.java:50-63), IV || ciphertext || tag per chunk with a
 * fresh SecureRandom IV (EncryptionChunkEnumeration.java:66-84, AesEncryptionProvider.java:66-71),
 * transformedChunkSize() == null when compressing, inner + 12 + 16 when only encrypting (:41-47, :82-84).
 * The C++ twin with the same logic is tiered-storage-for-apache-kafka_amd/host/tsxhost.cpp (tested); this file is the
 * JVM-side source a maintainer compiles - see INTEGRATION.md.
 */
package io.aiven.kafka.tieredstorage.gpu;

import java.nio.ByteBuffer;
import java.security.SecureRandom;
import java.util.ArrayDeque;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.List;
import java.util.NoSuchElementException;
import java.util.Objects;
import java.util.concurrent.CompletableFuture;
import java.util.concurrent.CompletionException;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Executors;

import io.aiven.kafka.tieredstorage.security.DataKeyAndAAD;
import io.aiven.kafka.tieredstorage.transform.TransformChunkEnumeration;

public class GpuTransformChunkEnumeration implements TransformChunkEnumeration {
    static final int IV_SIZE = 12;
    static final int TAG_SIZE = 16;

    private final TransformChunkEnumeration inner;
    private final boolean compress;
    private final DataKeyAndAAD keyAndAad;   // null: no encryption
    private final int batchChunks;
    private final SecureRandom random;
    private final int zstdProfile;
    private final int device;
    private final Integer transformedChunkSize;
    private final ArrayDeque<byte[]> ready = new ArrayDeque<>();
    private final boolean readAhead;
    /** The batch behind {@code ready}, being transformed by a helper; while it is set, only the helper touches {@code inner}. */
    private CompletableFuture<List<byte[]>> ahead;
    private boolean exhausted;

    /**
     * Long-lived helper threads (their pinned per-thread {@link TsxNative.Buffers} live as long as they do); as many as the
     * broker has upload threads by default (remote.log.manager.thread.pool.size = 10).
     */
    private static final ExecutorService HELPERS = Executors.newFixedThreadPool(
        Integer.getInteger("tsx.readahead.threads", 10), r -> {
            final Thread t = new Thread(r, "tsx-read-ahead");
            t.setDaemon(true);
            return t;
        });

    /**
     * @param zstdProfile TsxNative.ZSTD_PROFILE_1_5_6 (the libzstd inside the reference's zstd-jni 1.5.6-9, core/build.gradle:29)
     *                    or ZSTD_PROFILE_1_5_7; configuration key {@code gpu.zstd.profile} - INTEGRATION.md says what each
     *                    profile has been compared with
     * @param segmentHash any stable hash of the segment (e.g. of its RemoteLogSegmentId): its batches go to GPU
     *                    floorMod(segmentHash, deviceCount) so that the node's GPUs are all used and one segment stays on one
     * @param readAhead   while the consumer (TransformFinisher -> the uploader) drains batch k, batch k + 1 is already read
     *                    from {@code inner} and on the device: an upload thread keeps two batches in flight and its uploads
     *                    overlap the device.  Chunks, IVs and failures keep their order.  false: {@code inner} is read exactly
     *                    when the reference would read it.  (C++ twin, tested: tsx::GpuTransformChunkEnumeration, readAhead.)
     *                    ON in the constructor without this argument and as the default of {@code gpu.read.ahead}: the reference's default
     *                    of 10 upload threads offers the device 2560 chunks without it (0.41 of the device's rate: bench.py, end_to_end.broker),
     *                    5120 with it (0.81).  Costs one more batch of pinned host memory per upload thread (2.1 GiB for a 1 GiB segment).
     */
    public GpuTransformChunkEnumeration(final TransformChunkEnumeration inner, final boolean compress,
                                        final DataKeyAndAAD keyAndAad, final int batchChunks,
                                        final SecureRandom random, final int zstdProfile, final int segmentHash) {
        this(inner, compress, keyAndAad, batchChunks, random, zstdProfile, segmentHash, true);
    }

    public GpuTransformChunkEnumeration(final TransformChunkEnumeration inner, final boolean compress,
                                        final DataKeyAndAAD keyAndAad, final int batchChunks,
                                        final SecureRandom random, final int zstdProfile, final int segmentHash,
                                        final boolean readAhead) {
        this.inner = Objects.requireNonNull(inner, "inner cannot be null");
        this.compress = compress;
        this.keyAndAad = keyAndAad;
        this.batchChunks = batchChunks;
        this.random = random;
        if (zstdProfile != TsxNative.ZSTD_PROFILE_1_5_6 && zstdProfile != TsxNative.ZSTD_PROFILE_1_5_7) {
            throw new IllegalArgumentException("unknown Zstd profile " + zstdProfile);
        }
        this.zstdProfile = zstdProfile;
        this.readAhead = readAhead;
        this.device = Math.floorMod(segmentHash, TsxNative.deviceCount());
        // src and dst of a batch live in one direct ByteBuffer each (< 2 GiB)
        final long perChunk = TsxNative.transformedBound(inner.originalChunkSize(),
            (compress ? TsxNative.COMPRESS : 0) | (keyAndAad != null ? TsxNative.ENCRYPT : 0)) + 32;
        if (batchChunks < 1 || (long) batchChunks * perChunk >= Integer.MAX_VALUE - 64) {
            throw new IllegalArgumentException("batchChunks * chunk size must stay below 2 GiB, got " + batchChunks + " chunks of "
                + inner.originalChunkSize() + " bytes");
        }
        final Integer innerSize = inner.transformedChunkSize();
        if (compress || innerSize == null) {
            this.transformedChunkSize = null;
        } else {
            this.transformedChunkSize = keyAndAad != null ? innerSize + IV_SIZE + TAG_SIZE : innerSize;
        }
    }

    @Override
    public int originalChunkSize() {
        return inner.originalChunkSize();
    }

    @Override
    public Integer transformedChunkSize() {
        return transformedChunkSize;
    }

    @Override
    public boolean hasMoreElements() {
        fillBatchIfNeeded();
        return !ready.isEmpty();
    }

    @Override
    public byte[] nextElement() {
        fillBatchIfNeeded();
        if (ready.isEmpty()) {
            throw new NoSuchElementException();
        }
        return ready.poll();
    }

    private static long align16(final long v) {
        return (v + 15) & ~15L;
    }

    private void fillBatchIfNeeded() {
        if (!ready.isEmpty() || exhausted) {
            return;
        }
        final List<byte[]> batch;
        if (ahead != null) {
            final CompletableFuture<List<byte[]>> f = ahead;
            ahead = null;
            try {
                batch = f.join();              // the helper's failure surfaces here, where its first chunk is asked for
            } catch (final CompletionException e) {
                if (e.getCause() instanceof RuntimeException) {
                    throw (RuntimeException) e.getCause();
                }
                throw e;
            }
        } else {
            batch = transformNextBatch();
        }
        if (batch.isEmpty()) {
            exhausted = true;
            return;
        }
        ready.addAll(batch);
        if (readAhead) {
            ahead = CompletableFuture.supplyAsync(this::transformNextBatch, HELPERS);
        }
    }

    /** Up to batchChunks chunks of {@code inner} through the device; empty when {@code inner} is exhausted. */
    private List<byte[]> transformNextBatch() {
        final List<byte[]> out = new ArrayList<>();
        final List<byte[]> in = new ArrayList<>();
        while (in.size() < batchChunks && inner.hasMoreElements()) {
            in.add(inner.nextElement());
        }
        if (in.isEmpty()) {
            return out;
        }
        final int flags = (compress ? TsxNative.COMPRESS : 0) | (keyAndAad != null ? TsxNative.ENCRYPT : 0);
        // per-thread, reused, pinned (registered with the device): the compressor waves write every chunk's IV || C || TAG straight into the
        // dst buffer's slots (zero-copy output, DESIGN.md section 1) - a pageable buffer would send the batch through copy engines instead.
        // Footprint per thread: ~2.1 GiB at 256 x 4 MiB (source batch + bound-sized output slots), INTEGRATION.md section 4.
        final TsxNative.Buffers buffers = TsxNative.Buffers.get();
        final ByteBuffer descs = buffers.descs(in.size());
        long srcSize = 0;
        long dstSize = 0;
        final byte[] iv = new byte[IV_SIZE];
        for (int i = 0; i < in.size(); i++) {
            final int base = i * TsxNative.DESC_BYTES;
            final long cap = TsxNative.transformedBound(in.get(i).length, flags);
            descs.putLong(base + TsxNative.DESC_SRC_OFF, srcSize);
            descs.putLong(base + TsxNative.DESC_DST_OFF, dstSize);
            descs.putInt(base + TsxNative.DESC_SRC_LEN, in.get(i).length);
            descs.putInt(base + TsxNative.DESC_DST_CAP, (int) cap);
            descs.putInt(base + TsxNative.DESC_DST_LEN, 0);
            descs.putInt(base + TsxNative.DESC_STATUS, 0);
            if (keyAndAad != null) {
                random.nextBytes(iv);                      // the IV never comes from the device
                for (int k = 0; k < IV_SIZE; k++) {
                    descs.put(base + TsxNative.DESC_IV + k, iv[k]);
                }
            }
            srcSize += align16(in.get(i).length) + 16;
            dstSize += align16(cap) + 16;
        }
        final ByteBuffer src = buffers.src(srcSize + 16);
        final ByteBuffer dst = buffers.dst(dstSize + 16);
        for (int i = 0; i < in.size(); i++) {
            src.position((int) descs.getLong(i * TsxNative.DESC_BYTES + TsxNative.DESC_SRC_OFF));
            src.put(in.get(i));
        }
        TsxNative.setThreadDevice(device);
        final byte[] key = keyAndAad != null ? keyAndAad.dataKey.getEncoded() : null;   // a copy (SecretKeySpec.getEncoded clones)
        final int rc;
        try {
            rc = TsxNative.transformBatch(flags, key, keyAndAad != null ? keyAndAad.aad : null, zstdProfile, descs, in.size(), src, dst);
        } finally {
            if (key != null) {
                Arrays.fill(key, (byte) 0);            // the copy does not wait for the garbage collector
            }
            TsxNative.setThreadDevice(-1);             // the hint is this batch's: later ctx-less calls of the thread (a fetch) choose freely again
        }
        if (rc != TsxNative.OK) {
            throw new RuntimeException(TsxNative.strerror(rc));
        }
        for (int i = 0; i < in.size(); i++) {
            final int base = i * TsxNative.DESC_BYTES;
            final int status = descs.getInt(base + TsxNative.DESC_STATUS);
            if (status != TsxNative.OK) {
                throw new RuntimeException(TsxNative.strerror(status));   // as EncryptionChunkEnumeration.java:76-78
            }
            final byte[] chunk = new byte[descs.getInt(base + TsxNative.DESC_DST_LEN)];   // fresh array per chunk, owned by the caller
            dst.position((int) descs.getLong(base + TsxNative.DESC_DST_OFF));
            dst.get(chunk);
            out.add(chunk);
        }
        return out;
    }
}
