/*
 * Drop-in for DefaultChunkManager (core/.../fetch/DefaultChunkManager.java:36-70): same constructor shape, same
 * getChunk() contract (ranged fetch of chunk.range(), plain-text InputStream out), so ChunkManagerFactory and the
 * ChunkCache subclasses (which take a ChunkManager, ChunkManagerFactory.java:41-45) work unchanged.
 */
package io.aiven.kafka.tieredstorage.gpu;

import java.io.InputStream;
import java.util.List;
import java.util.Optional;

import io.aiven.kafka.tieredstorage.Chunk;
import io.aiven.kafka.tieredstorage.fetch.ChunkManager;
import io.aiven.kafka.tieredstorage.manifest.SegmentEncryptionMetadata;
import io.aiven.kafka.tieredstorage.manifest.SegmentManifest;
import io.aiven.kafka.tieredstorage.storage.ObjectFetcher;
import io.aiven.kafka.tieredstorage.storage.ObjectKey;
import io.aiven.kafka.tieredstorage.storage.StorageBackendException;
import io.aiven.kafka.tieredstorage.transform.BaseDetransformChunkEnumeration;
import io.aiven.kafka.tieredstorage.transform.DetransformChunkEnumeration;
import io.aiven.kafka.tieredstorage.transform.DetransformFinisher;

public class GpuChunkManager implements ChunkManager {
    private final ObjectFetcher fetcher;

    public GpuChunkManager(final ObjectFetcher fetcher) {
        this.fetcher = fetcher;
    }

    @Override
    public InputStream getChunk(final ObjectKey objectKey, final SegmentManifest manifest,
                                final int chunkId) throws StorageBackendException {
        final Chunk chunk = manifest.chunkIndex().chunks().get(chunkId);
        final InputStream chunkContent = fetcher.fetch(objectKey, chunk.range());
        DetransformChunkEnumeration detransformEnum = new BaseDetransformChunkEnumeration(chunkContent, List.of(chunk));
        final Optional<SegmentEncryptionMetadata> encryptionMetadata = manifest.encryption();
        if (encryptionMetadata.isPresent() || manifest.compression()) {
            detransformEnum = new GpuDetransformChunkEnumeration(detransformEnum, manifest.compression(),
                encryptionMetadata.orElse(null), chunk.originalSize, 1);
        }
        return new DetransformFinisher(detransformEnum).toInputStream();
    }
}
