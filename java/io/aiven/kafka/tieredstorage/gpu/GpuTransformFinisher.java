/*
 * The upload sink without per-chunk arrays (SURVEY section 8 f3, the JVM half).
 *
 * Replaces, at RemoteStorageManager.uploadSegmentLog (core/.../RemoteStorageManager.java:400-432), the pair
 *     transformation(...)  ->  TransformFinisher.newBuilder(transformEnum, size)...build()  ->  toInputStream() / chunkIndex()
 * (core/.../transform/TransformFinisher.java:48-151) when the chain runs on the GPU.  The reference's finisher hands the uploader a
 * SequenceInputStream over one ByteArrayInputStream per chunk (:101-110, :134-144); GpuTransformChunkEnumeration has to cut every
 * batch into fresh byte[] chunks for it (one allocation + one copy out of the direct buffer per chunk, then the uploader's own
 * partBuffer.put - storage/s3/.../S3MultiPartOutputStream.java:89-122).  Here a batch of chunks is transformed back to back into ONE
 * pinned direct buffer (TsxNative.transformBatchPacked = TSX_MEM_HOST_PACKED: the compressor waves write into it, the library packs
 * the slots down in place), the uploader's reads are served straight from that buffer, the chunk index is fed from the batch's
 * descriptor sizes, and the rate limit wraps the stream exactly as in the reference (TransformFinisher.java:146-151).
 *
 * Same observable behaviour as the reference's path: the object's bytes, the chunk index (addChunk for every chunk but the last,
 * finish for that one), "Chunk index was not built, was finisher used?" until the object has been read to its end, failures of a
 * chunk as RuntimeException when its batch is asked for.  The C++ twin with the same members and the same order of operations is
 * tsx::GpuTransformFinisher (tiered-storage-for-apache-kafka_amd/host/tsxhost.cpp), tested against the SequenceInputStream path
 * on the device (tests/host/host_tests.cpp, "GpuTransformFinisher"); this file is the JVM-side source a maintainer compiles -
 * INTEGRATION.md section 2b shows the call site.
 */
package io.aiven.kafka.tieredstorage.gpu;

import java.io.IOException;
import java.io.InputStream;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.security.SecureRandom;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.List;
import java.util.Objects;
import java.util.concurrent.CompletableFuture;
import java.util.concurrent.CompletionException;
import java.util.concurrent.ConcurrentLinkedDeque;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Executors;
import java.util.concurrent.atomic.AtomicInteger;

import io.aiven.kafka.tieredstorage.manifest.index.AbstractChunkIndexBuilder;
import io.aiven.kafka.tieredstorage.manifest.index.ChunkIndex;
import io.aiven.kafka.tieredstorage.manifest.index.FixedSizeChunkIndexBuilder;
import io.aiven.kafka.tieredstorage.manifest.index.VariableSizeChunkIndexBuilder;
import io.aiven.kafka.tieredstorage.security.DataKeyAndAAD;
import io.aiven.kafka.tieredstorage.transform.RateLimitedInputStream;
import io.aiven.kafka.tieredstorage.transform.TransformChunkEnumeration;

import io.github.bucket4j.Bucket;

public class GpuTransformFinisher {
    static final int IV_SIZE = 12;
    static final int TAG_SIZE = 16;

    /** One transformed batch: {@code object} holds its chunks back to back in [0, limit), {@code sizes} their transformed sizes. */
    private static final class PackedBatch {
        final ByteBuffer object;     // pinned, from PINNED; null for the empty batch that ends the object
        final int[] sizes;

        PackedBatch(final ByteBuffer object, final int[] sizes) {
            this.object = object;
            this.sizes = sizes;
        }
    }

    /**
     * Pinned direct buffers that outlive a finisher: registering ~1.1 GiB with the device costs milliseconds and must not be paid per
     * segment.  A finisher holds at most two (the batch being read, the batch being transformed); what comes back beyond
     * {@code tsx.packed.pool} buffers (default: two per read-ahead helper) is unpinned and left to the garbage collector.
     */
    static final class PinnedPool {
        private static final ConcurrentLinkedDeque<ByteBuffer> FREE = new ConcurrentLinkedDeque<>();
        private static final AtomicInteger KEPT = new AtomicInteger();
        private static final int MAX_KEPT = Integer.getInteger("tsx.packed.pool", 2 * Integer.getInteger("tsx.readahead.threads", 10));

        static ByteBuffer acquire(final long need) {
            if (need > Integer.MAX_VALUE - 64) {
                throw new IllegalArgumentException("batch of " + need + " bytes exceeds a direct ByteBuffer");
            }
            for (int tries = FREE.size(); tries > 0; tries--) {
                final ByteBuffer b = FREE.pollFirst();
                if (b == null) {
                    break;
                }
                if (b.capacity() >= need) {
                    KEPT.decrementAndGet();
                    b.clear();
                    return b;
                }
                FREE.addLast(b);       // too small for this batch: somebody else's, or replaced below when the pool is full
            }
            final ByteBuffer fresh = ByteBuffer.allocateDirect((int) Math.min(need + need / 16 + 64, (long) Integer.MAX_VALUE - 64))
                .order(ByteOrder.LITTLE_ENDIAN);
            TsxNative.hostRegister(fresh);          // best effort: an unpinned buffer still works (the library copies instead of writing in place)
            return fresh;
        }

        static void release(final ByteBuffer b) {
            if (b == null) {
                return;
            }
            if (KEPT.incrementAndGet() <= MAX_KEPT) {
                FREE.addFirst(b);
                return;
            }
            KEPT.decrementAndGet();
            TsxNative.hostUnregister(b);            // the pinning is undone before the garbage collector frees the memory
        }
    }

    /** The helpers that transform batch k + 1 while the uploader drains batch k (as in GpuTransformChunkEnumeration). */
    private static final ExecutorService HELPERS = Executors.newFixedThreadPool(
        Integer.getInteger("tsx.readahead.threads", 10), r -> {
            final Thread t = new Thread(r, "tsx-packed-read-ahead");
            t.setDaemon(true);
            return t;
        });

    private final TransformChunkEnumeration inner;
    private final boolean compress;
    private final DataKeyAndAAD keyAndAad;   // null: no encryption
    private final int batchChunks;
    private final SecureRandom random;
    private final int zstdProfile;
    private final int device;
    private final boolean readAhead;
    private final Bucket rateLimitingBucket;
    private final AbstractChunkIndexBuilder chunkIndexBuilder;

    private ByteBuffer current;              // the batch being read: position .. limit is what the uploader has not seen yet
    private CompletableFuture<PackedBatch> ahead;
    private int pendingSize = -1;            // size of the newest chunk: addChunk or finish, once it is known which
    private ChunkIndex chunkIndex;
    private boolean exhausted;

    /**
     * @param inner              the chunker over the segment file: BaseTransformChunkEnumeration(logSegmentInputStream, chunkSize),
     *                           RemoteStorageManager.java:438-441
     * @param originalFileSize   remoteLogSegmentMetadata.segmentSizeInBytes()
     * @param chunkingEnabled    false = TransformFinisher.Builder.withChunkingDisabled()
     * @param rateLimitingBucket null: no limit (TransformFinisher.java:146-151)
     * @param segmentHash        any stable hash of the segment: its batches go to GPU floorMod(segmentHash, deviceCount)
     * @param readAhead          batch k + 1 is read from {@code inner} and transformed while batch k is being uploaded
     */
    public GpuTransformFinisher(final TransformChunkEnumeration inner, final boolean compress, final DataKeyAndAAD keyAndAad,
                                final int batchChunks, final SecureRandom random, final int zstdProfile, final int segmentHash,
                                final int originalFileSize, final boolean chunkingEnabled, final Bucket rateLimitingBucket,
                                final boolean readAhead) {
        this.inner = Objects.requireNonNull(inner, "inner cannot be null");
        if (originalFileSize < 0) {
            throw new IllegalArgumentException("originalFileSize must be non-negative, " + originalFileSize + " given");
        }
        if (zstdProfile != TsxNative.ZSTD_PROFILE_1_5_6 && zstdProfile != TsxNative.ZSTD_PROFILE_1_5_7) {
            throw new IllegalArgumentException("unknown Zstd profile " + zstdProfile);
        }
        this.compress = compress;
        this.keyAndAad = keyAndAad;
        this.batchChunks = batchChunks;
        this.random = random;
        this.zstdProfile = zstdProfile;
        this.readAhead = readAhead;
        this.rateLimitingBucket = rateLimitingBucket;
        this.device = Math.floorMod(segmentHash, TsxNative.deviceCount());
        final int flags = flags();
        // one bound-sized slot per chunk of a batch must fit one direct ByteBuffer (< 2 GiB); without chunking the one chunk is the whole file
        final long largestChunk = chunkingEnabled ? inner.originalChunkSize() : originalFileSize;
        final long slot = align64(TsxNative.transformedBound(largestChunk, flags));
        if (batchChunks < 1 || (long) batchChunks * (slot + 32) >= Integer.MAX_VALUE - 64) {
            throw new IllegalArgumentException("batchChunks * chunk size must stay below 2 GiB, got " + batchChunks + " chunks of "
                + inner.originalChunkSize() + " bytes");
        }
        // the index builder of TransformFinisher.java:64-93: variable sizes when compressing, else inner (+ IV + tag)
        final int originalChunkSize = chunkingEnabled ? inner.originalChunkSize() : originalFileSize;
        final Integer innerSize = inner.transformedChunkSize();
        final Integer transformedChunkSize;
        if (compress || innerSize == null) {
            transformedChunkSize = null;
        } else {
            transformedChunkSize = keyAndAad != null ? innerSize + IV_SIZE + TAG_SIZE : innerSize;
        }
        this.chunkIndexBuilder = transformedChunkSize == null
            ? new VariableSizeChunkIndexBuilder(originalChunkSize, originalFileSize)
            : new FixedSizeChunkIndexBuilder(originalChunkSize, originalFileSize, transformedChunkSize);
    }

    private int flags() {
        return (compress ? TsxNative.COMPRESS : 0) | (keyAndAad != null ? TsxNative.ENCRYPT : 0);
    }

    private static long align16(final long v) {
        return (v + 15) & ~15L;
    }

    private static long align64(final long v) {
        return (v + 63) & ~63L;
    }

    /** TransformFinisher.chunkIndex(): complete once the object has been read to its end. */
    public ChunkIndex chunkIndex() {
        if (chunkIndex == null) {
            throw new IllegalStateException("Chunk index was not built, was finisher used?");
        }
        return chunkIndex;
    }

    /** TransformFinisher.toInputStream(): the transformed object, rate limited when a bucket was given. */
    public InputStream toInputStream() {
        final InputStream packed = new PackedObjectStream();
        if (rateLimitingBucket == null) {
            return packed;
        }
        return new RateLimitedInputStream(packed, rateLimitingBucket);
    }

    /**
     * For a sink that takes ByteBuffers (an uploader with part buffers of its own): what is left of the current batch as a read-only
     * view of the pinned buffer - no copy at all - or null at the object's end.  The view is valid until the next call.
     */
    public ByteBuffer nextPackedBatch() {
        if ((current == null || !current.hasRemaining()) && !nextBatch()) {
            return null;
        }
        final ByteBuffer view = current.asReadOnlyBuffer();
        current.position(current.limit());
        return view;
    }

    /**
     * For a sink with a part buffer of its own (S3MultiPartOutputStream.java:89-122 does partBuffer.put(inputBuffer.slice())): fills
     * {@code part} from the object, across batches, one bulk put per batch; returns the bytes written - less than part.remaining()
     * only at the object's end.
     */
    public int fillPart(final ByteBuffer part) {
        int written = 0;
        while (part.hasRemaining()) {
            if ((current == null || !current.hasRemaining()) && !nextBatch()) {
                break;
            }
            final int m = Math.min(part.remaining(), current.remaining());
            final ByteBuffer piece = current.duplicate();
            piece.limit(piece.position() + m);
            part.put(piece);
            current.position(current.position() + m);
            written += m;
        }
        return written;
    }

    /** Releases the pinned buffers (also done when the object's end is reached); the finisher cannot be used afterwards. */
    public void close() {
        if (ahead != null) {
            try {
                final PackedBatch b = ahead.join();
                PinnedPool.release(b.object);
            } catch (final CompletionException e) {
                // the helper's failure has nobody left to go to
            }
            ahead = null;
        }
        PinnedPool.release(current);
        current = null;
        exhausted = true;
    }

    /**
     * The next packed batch becomes the one being read.  Sizes go to the index builder in order, the newest one held back: the
     * reference calls addChunk for every chunk but the object's last and finish for that one (TransformFinisher.java:101-110), and
     * which chunk is the last is only known when the batch behind it comes back empty.
     */
    private boolean nextBatch() {
        if (exhausted) {
            return false;
        }
        PinnedPool.release(current);
        current = null;
        final PackedBatch batch;
        if (ahead != null) {
            final CompletableFuture<PackedBatch> f = ahead;
            ahead = null;
            try {
                batch = f.join();              // the helper's failure surfaces here, where its first byte is asked for
            } catch (final CompletionException e) {
                exhausted = true;
                if (e.getCause() instanceof RuntimeException) {
                    throw (RuntimeException) e.getCause();
                }
                throw e;
            }
        } else {
            batch = transformNextBatchPacked();
        }
        if (batch.sizes.length == 0) {
            exhausted = true;
            if (pendingSize >= 0) {
                chunkIndex = chunkIndexBuilder.finish(pendingSize);
                pendingSize = -1;
            }
            return false;
        }
        for (final int size : batch.sizes) {
            if (pendingSize >= 0) {
                chunkIndexBuilder.addChunk(pendingSize);
            }
            pendingSize = size;
        }
        current = batch.object;
        if (readAhead) {
            ahead = CompletableFuture.supplyAsync(this::transformNextBatchPacked, HELPERS);
        }
        return true;
    }

    /** Up to batchChunks chunks of {@code inner} through the device, back to back in one pinned buffer; no sizes: {@code inner} is exhausted. */
    private PackedBatch transformNextBatchPacked() {
        final List<byte[]> in = new ArrayList<>();
        while (in.size() < batchChunks && inner.hasMoreElements()) {
            in.add(inner.nextElement());
        }
        if (in.isEmpty()) {
            return new PackedBatch(null, new int[0]);
        }
        final int flags = flags();
        final TsxNative.Buffers buffers = TsxNative.Buffers.get();        // source staging + descriptors: per thread, pinned, reused
        final ByteBuffer descs = buffers.descs(in.size());
        long srcSize = 0;
        long maxLen = 0;
        final byte[] iv = new byte[IV_SIZE];
        for (int i = 0; i < in.size(); i++) {
            final int base = i * TsxNative.DESC_BYTES;
            descs.putLong(base + TsxNative.DESC_SRC_OFF, srcSize);
            descs.putLong(base + TsxNative.DESC_DST_OFF, 0);             // TSX_MEM_HOST_PACKED: the library says where the chunk landed
            descs.putInt(base + TsxNative.DESC_SRC_LEN, in.get(i).length);
            descs.putInt(base + TsxNative.DESC_DST_CAP, 0);
            descs.putInt(base + TsxNative.DESC_DST_LEN, 0);
            descs.putInt(base + TsxNative.DESC_STATUS, 0);
            if (keyAndAad != null) {
                random.nextBytes(iv);                      // the IV never comes from the device
                for (int k = 0; k < IV_SIZE; k++) {
                    descs.put(base + TsxNative.DESC_IV + k, iv[k]);
                }
            }
            srcSize += align16(in.get(i).length) + 16;
            maxLen = Math.max(maxLen, in.get(i).length);
        }
        final ByteBuffer src = buffers.src(srcSize + 16);
        for (int i = 0; i < in.size(); i++) {
            src.position((int) descs.getLong(i * TsxNative.DESC_BYTES + TsxNative.DESC_SRC_OFF));
            src.put(in.get(i));
        }
        // room for one bound-sized slot per chunk: the waves fill the slots in place (zero-copy output) and the library packs them down
        final ByteBuffer object = PinnedPool.acquire((long) in.size() * align64(TsxNative.transformedBound(maxLen, flags)) + 64);
        TsxNative.setThreadDevice(device);
        final byte[] key = keyAndAad != null ? keyAndAad.dataKey.getEncoded() : null;   // a copy (SecretKeySpec.getEncoded clones)
        final int rc;
        try {
            rc = TsxNative.transformBatchPacked(flags, key, keyAndAad != null ? keyAndAad.aad : null, zstdProfile, descs, in.size(), src, object);
        } finally {
            if (key != null) {
                Arrays.fill(key, (byte) 0);            // the copy does not wait for the garbage collector
            }
            TsxNative.setThreadDevice(-1);
        }
        if (rc != TsxNative.OK) {
            PinnedPool.release(object);
            throw new RuntimeException(TsxNative.strerror(rc));
        }
        final int[] sizes = new int[in.size()];
        long at = 0;
        for (int i = 0; i < in.size(); i++) {
            final int base = i * TsxNative.DESC_BYTES;
            final int status = descs.getInt(base + TsxNative.DESC_STATUS);
            if (status != TsxNative.OK) {
                PinnedPool.release(object);
                throw new RuntimeException(TsxNative.strerror(status));   // as EncryptionChunkEnumeration.java:76-78
            }
            sizes[i] = descs.getInt(base + TsxNative.DESC_DST_LEN);
            if (descs.getLong(base + TsxNative.DESC_DST_OFF) != at) {
                PinnedPool.release(object);
                throw new IllegalStateException("packed batch is not contiguous at chunk " + i);
            }
            at += sizes[i];
        }
        object.position(0);
        object.limit((int) at);
        return new PackedBatch(object, sizes);
    }

    /** new SequenceInputStream(transformFinisher), without the chunk arrays: a read never crosses a batch. */
    private final class PackedObjectStream extends InputStream {
        private final byte[] one = new byte[1];

        @Override
        public int read() throws IOException {
            return read(one, 0, 1) < 0 ? -1 : one[0] & 0xFF;
        }

        @Override
        public int read(final byte[] b, final int off, final int len) throws IOException {
            Objects.checkFromIndexSize(off, len, b.length);
            if (len == 0) {
                return 0;
            }
            while (current == null || !current.hasRemaining()) {
                if (!nextBatch()) {
                    return -1;
                }
            }
            final int m = Math.min(len, current.remaining());
            current.get(b, off, m);
            return m;
        }

        @Override
        public int available() {
            return current == null ? 0 : current.remaining();
        }

        @Override
        public void close() {
            GpuTransformFinisher.this.close();
        }
    }
}
