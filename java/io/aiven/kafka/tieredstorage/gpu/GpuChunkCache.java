/*
 * Chunk cache for a GpuChunkManager underneath, selectable exactly like the reference's caches: a subclass of
 * fetch/cache/ChunkCache with the (ChunkManager) constructor and configure(Map) that ChunkManagerFactory.java:39-46 calls, so
 *     fetch.chunk.cache.class = io.aiven.kafka.tieredstorage.gpu.GpuChunkCache
 * is all it takes (keys under fetch.chunk.cache.: size, prefetch.max.size, get.timeout.ms, thread.pool.size as in ChunkCacheConfig,
 * plus gpu.coalesce.wait.us).  The factory hands every cache a DefaultChunkManager; this one keeps only its ObjectFetcher and
 * puts a GpuChunkManager in its place (a GpuChunkManager passed in directly is used as it is).
 *
 * What differs from ChunkCache.getChunk (ChunkCache.java:76-129) / startPrefetching (:159-184): the reference turns a prefetch
 * window of k chunks into k single-chunk tasks (one ranged fetch and one detransform each); through a GPU that is k launches.
 * Here the requested chunk and the not-yet-cached part of its window become ONE GpuChunkManager.getChunks call - one ranged fetch,
 * one device batch - and concurrent misses on the next chunks of the same object that arrive within a short bounded wait
 * (gpu.coalesce.wait.us, far below get.timeout.ms) join the batch that is about to leave.  Kept from the reference: nothing beyond
 * the configured window is fetched, a cached chunk is handed out as a fresh stream over its bytes, every load runs on the cache's
 * executor and a waiter gives up after get.timeout.ms with a RuntimeException around the TimeoutException, a chunk that fails
 * (tag mismatch, corrupt frame) fails only the callers of that chunk (the window is retried chunk by chunk), weight-bounded eviction.
 * The executor is a fixed pool of long-lived daemon threads: each owns pinned direct buffers (TsxNative.Buffers) that must not be
 * freed by the collector while still registered with the device runtime, so the threads never time out and release their buffers
 * when the pool is closed.
 * The C++ twin with the same logic is tested (tiered-storage-for-apache-kafka_amd/host/tsxhost.cpp, tsx::GpuChunkCache;
 * tests/host/host_tests.cpp "GpuChunkCache"); this file is the JVM-side source a maintainer compiles - see INTEGRATION.md.
 */
package io.aiven.kafka.tieredstorage.gpu;

import java.io.ByteArrayInputStream;
import java.io.IOException;
import java.io.InputStream;
import java.lang.reflect.Field;
import java.util.ArrayList;
import java.util.HashMap;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import java.util.concurrent.CompletableFuture;
import java.util.concurrent.ExecutionException;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Executors;
import java.util.concurrent.RejectedExecutionException;
import java.util.concurrent.TimeUnit;
import java.util.concurrent.atomic.AtomicInteger;
import java.util.concurrent.TimeoutException;

import com.github.benmanes.caffeine.cache.RemovalListener;
import com.github.benmanes.caffeine.cache.Weigher;

import io.aiven.kafka.tieredstorage.Chunk;
import io.aiven.kafka.tieredstorage.config.ChunkCacheConfig;
import io.aiven.kafka.tieredstorage.fetch.ChunkKey;
import io.aiven.kafka.tieredstorage.fetch.ChunkManager;
import io.aiven.kafka.tieredstorage.fetch.DefaultChunkManager;
import io.aiven.kafka.tieredstorage.fetch.cache.ChunkCache;
import io.aiven.kafka.tieredstorage.manifest.SegmentManifest;
import io.aiven.kafka.tieredstorage.storage.BytesRange;
import io.aiven.kafka.tieredstorage.storage.ObjectFetcher;
import io.aiven.kafka.tieredstorage.storage.ObjectKey;
import io.aiven.kafka.tieredstorage.storage.StorageBackendException;

public class GpuChunkCache extends ChunkCache<byte[]> implements AutoCloseable {
    private static final class Batch {
        final ObjectKey object;
        final int first;
        boolean open = true;                                   // still accepts the chunks right behind its end
        final List<CompletableFuture<byte[]>> slots = new ArrayList<>();

        Batch(final ObjectKey object, final int first) {
            this.object = object;
            this.first = first;
        }
    }

    static final String COALESCE_WAIT_US_CONFIG = "gpu.coalesce.wait.us";

    private final GpuChunkManager manager;
    private int prefetchingSize;
    private long maxBytes = Long.MAX_VALUE;
    private long getTimeoutMs = 10_000;
    private long coalesceWaitMicros = 200;
    private long retentionMs = Long.MAX_VALUE;
    /** last access of every cached chunk (retention.ms); same keys as {@code cached}. */
    private final Map<ChunkKey, Long> touched = new HashMap<>();
    private ExecutorService executor;

    private final Object lock = new Object();
    /** access-ordered: iteration starts at the least recently used entry. */
    private final LinkedHashMap<ChunkKey, byte[]> cached = new LinkedHashMap<>(64, 0.75f, true);
    private final Map<ChunkKey, CompletableFuture<byte[]>> pending = new HashMap<>();
    private final Map<String, Batch> openBatch = new HashMap<>();
    private long bytes;

    /** The constructor ChunkManagerFactory looks up (ChunkManagerFactory.java:41-43); configure(Map) follows. */
    public GpuChunkCache(final ChunkManager chunkManager) {
        super(chunkManager);
        this.manager = chunkManager instanceof GpuChunkManager
            ? (GpuChunkManager) chunkManager
            : new GpuChunkManager(fetcherOf(chunkManager));
    }

    /** For embedders that build the chain themselves (and the C++ twin's test cases). */
    public GpuChunkCache(final GpuChunkManager manager, final int prefetchingSize, final long maxBytes,
                         final long getTimeoutMs, final long coalesceWaitMicros) {
        super(manager);
        this.manager = manager;
        this.prefetchingSize = prefetchingSize;
        this.maxBytes = maxBytes;
        this.getTimeoutMs = getTimeoutMs;
        this.coalesceWaitMicros = coalesceWaitMicros;
        this.executor = newExecutor(Runtime.getRuntime().availableProcessors());
    }

    /** The factory's DefaultChunkManager owns the ObjectFetcher this cache fetches ranges with; nothing else of it is used. */
    private static ObjectFetcher fetcherOf(final ChunkManager chunkManager) {
        if (!(chunkManager instanceof DefaultChunkManager)) {
            throw new IllegalArgumentException("GpuChunkCache wraps a DefaultChunkManager or a GpuChunkManager, not "
                + chunkManager.getClass().getName());
        }
        try {
            final Field f = DefaultChunkManager.class.getDeclaredField("fetcher");
            f.setAccessible(true);
            return (ObjectFetcher) f.get(chunkManager);
        } catch (final ReflectiveOperationException e) {
            throw new IllegalStateException("DefaultChunkManager.fetcher is not reachable: pass a GpuChunkManager instead", e);
        }
    }

    @Override
    public void configure(final Map<String, ?> configs) {
        final ChunkCacheConfig config = new ChunkCacheConfig(configs);    // same keys, defaults and validation as the reference's caches
        this.prefetchingSize = config.cachePrefetchingSize();
        this.maxBytes = config.cacheSize().orElse(Long.MAX_VALUE);       // size = -1: unbounded
        this.getTimeoutMs = config.getTimeout().toMillis();
        final Object wait = configs.get(COALESCE_WAIT_US_CONFIG);
        if (wait != null) {
            this.coalesceWaitMicros = Long.parseLong(wait.toString().trim());
            if (this.coalesceWaitMicros < 0 || this.coalesceWaitMicros > 1000L * this.getTimeoutMs / 2) {
                throw new IllegalArgumentException(COALESCE_WAIT_US_CONFIG + " must lie in [0, get.timeout.ms / 2]");
            }
        }
        // retention.ms: the reference's caches expire an entry that has not been read for that long (Caffeine expireAfterAccess,
        // ChunkCache.java:146-150); here entries carry their last-access time and are dropped on the next insert / lookup after it
        this.retentionMs = config.cacheRetention().map(java.time.Duration::toMillis).orElse(Long.MAX_VALUE);
        final ExecutorService old = this.executor;
        this.executor = newExecutor(config.threadPoolSize().orElse(Runtime.getRuntime().availableProcessors()));
        if (old != null) {
            old.shutdown();                                            // a re-configured cache does not leak its first pool's threads
        }
    }

    private static ExecutorService newExecutor(final int threads) {
        final AtomicInteger id = new AtomicInteger();
        // a fixed pool: core threads never time out, so a thread's pinned buffers live as long as the thread
        return Executors.newFixedThreadPool(Math.max(1, threads), r -> {
            final Thread t = new Thread(() -> {
                try {
                    r.run();
                } finally {
                    TsxNative.Buffers.release();                         // the worker ends (pool shut down): unpin before the collector frees
                }
            }, "gpu-chunk-cache-" + id.getAndIncrement());
            t.setDaemon(true);
            return t;
        });
    }

    @Override
    public void close() {
        if (executor != null) {
            executor.shutdown();
        }
    }

    // ChunkCache's Caffeine-facing hooks: this cache keeps its own weight-bounded LRU of byte[] (below), they only state the types
    @Override
    public InputStream cachedChunkToInputStream(final byte[] cachedChunk) {
        return new ByteArrayInputStream(cachedChunk);
    }

    @Override
    public byte[] cacheChunk(final ChunkKey chunkKey, final InputStream chunk) throws IOException {
        try (chunk) {
            return chunk.readAllBytes();
        }
    }

    @Override
    public RemovalListener<ChunkKey, byte[]> removalListener() {
        return (key, content, cause) -> { };
    }

    @Override
    public Weigher<ChunkKey, byte[]> weigher() {
        return (key, value) -> value.length;
    }

    private void insert(final ChunkKey key, final byte[] value) {          // under lock
        final long now = System.currentTimeMillis();
        final byte[] old = cached.put(key, value);
        touched.put(key, now);
        bytes += value.length - (old != null ? old.length : 0);
        final var it = cached.entrySet().iterator();                       // least recently used first
        while (it.hasNext()) {
            final var eldest = it.next();
            if (eldest.getKey().equals(key)) {
                continue;
            }
            final boolean expired = retentionMs != Long.MAX_VALUE && now - touched.getOrDefault(eldest.getKey(), now) > retentionMs;
            if (!expired && !(bytes > maxBytes && cached.size() > 1)) {
                break;                                                     // access order = age order: nothing behind it is older
            }
            bytes -= eldest.getValue().length;
            touched.remove(eldest.getKey());
            it.remove();
        }
    }

    /** A cached chunk, unless retention.ms has passed since its last read (then it is dropped, as an expired Caffeine entry is). */
    private byte[] lookup(final ChunkKey key) {                           // under lock
        final byte[] hit = cached.get(key);
        if (hit == null) {
            return null;
        }
        final long now = System.currentTimeMillis();
        if (retentionMs != Long.MAX_VALUE && now - touched.getOrDefault(key, now) > retentionMs) {
            bytes -= hit.length;
            cached.remove(key);
            touched.remove(key);
            return null;
        }
        touched.put(key, now);
        return hit;
    }

    /**
     * One ranged fetch + one device batch; on failure every chunk on its own, so that only the bad one fails.  Runs on the executor:
     * EVERY Throwable is captured (an OutOfMemoryError "Direct buffer memory" of the per-thread buffers, an UnsatisfiedLinkError of
     * the JNI pair ...) the way CompletableFuture.supplyAsync does for the reference (ChunkCache.java:76-129), and whatever happens no
     * slot stays open and no key stays in {@code pending}: later getChunk calls for these ids must start a new load, not wait
     * get.timeout.ms for a task that died.
     */
    private void runBatch(final Batch batch, final SegmentManifest manifest) {
        final int count = batch.slots.size();
        Throwable fatal = null;
        try {
            List<byte[]> got = null;
            Throwable batchError = null;
            try {
                got = manager.getChunks(batch.object, manifest, batch.first, count);
            } catch (final Throwable e) {
                batchError = e;
            }
            for (int i = 0; i < count; i++) {
                final ChunkKey key = new ChunkKey(batch.object.value(), batch.first + i);
                byte[] value = null;
                Throwable error = null;
                if (batchError == null) {
                    value = got.get(i);
                } else if (count == 1 || batchError instanceof Error) {
                    error = batchError;                            // an Error is not a property of one chunk: no chunk-by-chunk retry
                } else {
                    try {
                        value = manager.getChunks(batch.object, manifest, batch.first + i, 1).get(0);
                    } catch (final Throwable e) {
                        error = e;
                    }
                }
                synchronized (lock) {
                    if (error == null) {
                        insert(key, value);
                    }
                    pending.remove(key);
                }
                if (error == null) {
                    batch.slots.get(i).complete(value);
                } else {
                    batch.slots.get(i).completeExceptionally(error);
                }
            }
        } catch (final Throwable e) {
            fatal = e;
        } finally {
            abandon(batch, fatal != null ? fatal : new IllegalStateException("chunk load ended without a result"));
        }
    }

    /** Fails every slot of the batch that has no result yet and forgets its keys (no-op for slots that are done). */
    private void abandon(final Batch batch, final Throwable cause) {
        for (int i = 0; i < batch.slots.size(); i++) {
            final CompletableFuture<byte[]> slot = batch.slots.get(i);
            if (!slot.isDone()) {
                synchronized (lock) {
                    pending.remove(new ChunkKey(batch.object.value(), batch.first + i), slot);
                }
                slot.completeExceptionally(cause);
            }
        }
    }

    /** executor.execute that cannot leave a batch behind: a closed (or saturated) executor fails the batch's waiters at once. */
    private void submit(final Batch batch, final SegmentManifest manifest) {
        try {
            executor.execute(() -> runBatch(batch, manifest));
        } catch (final RejectedExecutionException e) {
            abandon(batch, e);
        }
    }

    @Override
    public InputStream getChunk(final ObjectKey objectKey, final SegmentManifest manifest, final int chunkId)
        throws StorageBackendException {
        final List<Chunk> all = manifest.chunkIndex().chunks();
        // the window this call may touch: the chunk itself + ChunkCache.startPrefetching's range behind it (never more)
        int last = chunkId;
        if (prefetchingSize > 0) {
            final Chunk current = all.get(chunkId);
            final int start = current.originalPosition + current.originalSize;
            final BytesRange range = Integer.MAX_VALUE - start < prefetchingSize
                ? BytesRange.of(start, Integer.MAX_VALUE)
                : BytesRange.ofFromPositionAndSize(start, prefetchingSize);
            for (final Chunk c : manifest.chunkIndex().chunksForRange(range)) {
                last = Math.max(last, c.id);
            }
        }
        CompletableFuture<byte[]> mine = null;
        Batch lead = null;
        synchronized (lock) {
            final ChunkKey key = new ChunkKey(objectKey.value(), chunkId);
            final byte[] hit = lookup(key);
            if (hit == null) {
                mine = pending.get(key);
            }
            // what is neither cached nor on its way, as runs of consecutive ids
            int i = chunkId;
            while (i <= last) {
                while (i <= last && known(objectKey, i)) {
                    i++;
                }
                if (i > last) {
                    break;
                }
                int j = i;
                while (j <= last && !known(objectKey, j)) {
                    j++;
                }
                final Batch open = openBatch.get(objectKey.value());
                final boolean joins = open != null && open.open && open.first + open.slots.size() == i;
                final Batch batch = joins ? open : new Batch(objectKey, i);
                for (int c = i; c < j; c++) {
                    final CompletableFuture<byte[]> slot = new CompletableFuture<>();
                    batch.slots.add(slot);
                    pending.put(new ChunkKey(objectKey.value(), c), slot);
                }
                if (i <= chunkId && chunkId < j) {
                    mine = pending.get(key);
                }
                if (!joins) {
                    if (i <= chunkId && chunkId < j && lead == null) {
                        lead = batch;
                        openBatch.put(objectKey.value(), batch);
                    } else {
                        batch.open = false;                    // pure prefetch: nobody waits for it here
                        submit(batch, manifest);
                    }
                }
                i = j;
            }
            if (hit != null) {
                return new ByteArrayInputStream(hit);
            }
        }
        if (lead != null) {
            if (coalesceWaitMicros > 0) {
                try {
                    TimeUnit.MICROSECONDS.sleep(coalesceWaitMicros);   // bounded: microseconds against get.timeout.ms
                } catch (final InterruptedException e) {
                    Thread.currentThread().interrupt();
                }
            }
            synchronized (lock) {
                lead.open = false;
                openBatch.remove(objectKey.value(), lead);
            }
            final Batch leaving = lead;
            submit(leaving, manifest);                                 // on the executor like every load: get.timeout.ms bounds the leader too
        }
        try {
            return new ByteArrayInputStream(mine.get(getTimeoutMs, TimeUnit.MILLISECONDS));
        } catch (final ExecutionException e) {
            if (e.getCause() instanceof StorageBackendException) {
                throw (StorageBackendException) e.getCause();      // unwrapped like ChunkCache.java:112-124
            }
            throw new RuntimeException(e.getCause() != null ? e.getCause() : e);
        } catch (final InterruptedException | TimeoutException e) {
            throw new RuntimeException(e);
        }
    }

    private boolean known(final ObjectKey objectKey, final int id) {       // under lock
        final ChunkKey k = new ChunkKey(objectKey.value(), id);
        return cached.containsKey(k) || pending.containsKey(k);
    }
}
