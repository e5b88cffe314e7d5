// tsxhost — host side of the chunk-transform path, above the C ABI of libtsxform.so (include/tsxform.h).
//
// The reference's host code is Java; no JDK exists in this image, so the same interfaces are provided here in C++ with
// the reference's names, argument meaning and error behaviour (java/ holds the Java classes + JNI shim a maintainer would
// compile, INTEGRATION.md shows where they plug in).  Mirrored reference types (core/src/main/java/io/aiven/kafka/tieredstorage/):
//   Chunk.java:21-66                                   -> tsx::Chunk
//   manifest/index/AbstractChunkIndex.java:30-128      -> tsx::ChunkIndex (+ FixedSizeChunkIndex.java:30-91, VariableSizeChunkIndex.java:30-87)
//   manifest/index/AbstractChunkIndexBuilder.java:25-96, FixedSizeChunkIndexBuilder.java:25-44, VariableSizeChunkIndexBuilder.java:25-42
//   manifest/index/serde/ChunkSizesBinaryCodec.java:104-202, TransformedChunksSerializer.java:28-53, TransformedChunksDeserializer.java:27-48
//   transform/TransformChunkEnumeration.java:28-42, BaseTransformChunkEnumeration.java:30-97, TransformFinisher.java:40-151
//   transform/DetransformChunkEnumeration.java:28, BaseDetransformChunkEnumeration.java:39-115, DetransformFinisher.java:30-54
//   transform/CompressionChunkEnumeration.java + EncryptionChunkEnumeration.java  -> fused tsx::GpuTransformChunkEnumeration
//   transform/DecryptionChunkEnumeration.java + DecompressionChunkEnumeration.java -> fused tsx::GpuDetransformChunkEnumeration
//   fetch/ChunkManager.java:26-31, fetch/DefaultChunkManager.java:36-70               -> tsx::ChunkManager, tsx::GpuChunkManager
// Java exceptions map to: IllegalArgumentException -> std::invalid_argument, IllegalStateException -> std::logic_error,
// NoSuchElementException -> std::out_of_range, RuntimeException -> std::runtime_error (same messages).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <future>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tsxform.h"

namespace tsx {

using Bytes = std::vector<uint8_t>;

// ---- storage/BytesRange + Chunk ------------------------------------------------------------------------
struct BytesRange {                       // inclusive [from, to]
    int from, to;
    static BytesRange ofFromPositionAndSize(int position, int size) { return BytesRange{position, position + size - 1}; }
    int size() const { return to - from + 1; }
};

struct Chunk {
    int id, originalPosition, originalSize, transformedPosition, transformedSize;
    BytesRange range() const { return BytesRange::ofFromPositionAndSize(transformedPosition, transformedSize); }
    bool operator==(const Chunk& o) const {
        return id == o.id && originalPosition == o.originalPosition && originalSize == o.originalSize &&
               transformedPosition == o.transformedPosition && transformedSize == o.transformedSize;
    }
};

// ---- chunk index ---------------------------------------------------------------------------------------
class ChunkIndex {
public:
    virtual ~ChunkIndex() = default;
    const std::vector<Chunk>& chunks() const { return chunks_; }
    // nullopt at / after the end of the original file; throws std::invalid_argument on a negative offset
    std::optional<Chunk> findChunkForOriginalOffset(int offset) const;
    std::vector<Chunk> chunksForRange(BytesRange range) const;
    int originalChunkSize() const { return originalChunkSize_; }
    int originalFileSize() const { return originalFileSize_; }
    virtual bool isFixed() const = 0;
    virtual int transformedChunkSize(int chunkI) const = 0;

protected:
    ChunkIndex(int originalChunkSize, int originalFileSize, int finalTransformedChunkSize, int chunkCount);
    void materializeChunks();
    int originalChunkSizeOf(int chunkI) const;
    int originalChunkSize_, originalFileSize_, finalTransformedChunkSize_, chunkCount_;
    std::vector<Chunk> chunks_;
};

class FixedSizeChunkIndex : public ChunkIndex {
public:
    FixedSizeChunkIndex(int originalChunkSize, int originalFileSize, int transformedChunkSize, int finalTransformedChunkSize);
    bool isFixed() const override { return true; }
    int transformedChunkSize(int chunkI) const override { return chunkI == chunkCount_ - 1 ? finalTransformedChunkSize_ : transformedChunkSize_; }
    int transformedChunkSize() const { return transformedChunkSize_; }
    int finalTransformedChunkSize() const { return finalTransformedChunkSize_; }

private:
    int transformedChunkSize_;
};

class VariableSizeChunkIndex : public ChunkIndex {
public:
    VariableSizeChunkIndex(int originalChunkSize, int originalFileSize, std::vector<int> transformedChunks);
    bool isFixed() const override { return false; }
    int transformedChunkSize(int chunkI) const override { return transformedChunks_[(size_t)chunkI]; }
    const std::vector<int>& transformedChunks() const { return transformedChunks_; }

private:
    std::vector<int> transformedChunks_;
};

class AbstractChunkIndexBuilder {
public:
    virtual ~AbstractChunkIndexBuilder() = default;
    void addChunk(int transformedChunkSize);                          // every chunk but the last
    std::shared_ptr<ChunkIndex> finish(int finalTransformedChunkSize);   // the last chunk

protected:
    AbstractChunkIndexBuilder(int originalChunkSize, int originalFileSize);
    virtual void addChunk0(int transformedChunkSize) = 0;
    virtual std::shared_ptr<ChunkIndex> finish0(int finalTransformedChunkSize) = 0;
    static void checkSize(int size, const char* name);
    int remainOfOriginalFileSize() const { return originalFileSize_ - chunksAdded_ * originalChunkSize_; }
    int originalChunkSize_, originalFileSize_, chunksAdded_ = 0;
    bool finished_ = false;
};

class FixedSizeChunkIndexBuilder : public AbstractChunkIndexBuilder {
public:
    FixedSizeChunkIndexBuilder(int originalChunkSize, int originalFileSize, int transformedChunkSize);

protected:
    void addChunk0(int transformedChunkSize) override;
    std::shared_ptr<ChunkIndex> finish0(int finalTransformedChunkSize) override;

private:
    int transformedChunkSize_;
};

class VariableSizeChunkIndexBuilder : public AbstractChunkIndexBuilder {
public:
    VariableSizeChunkIndexBuilder(int originalChunkSize, int originalFileSize) : AbstractChunkIndexBuilder(originalChunkSize, originalFileSize) {}

protected:
    void addChunk0(int transformedChunkSize) override { transformedChunks_.push_back(transformedChunkSize); }
    std::shared_ptr<ChunkIndex> finish0(int finalTransformedChunkSize) override;

private:
    std::vector<int> transformedChunks_;
};

struct ChunkSizesBinaryCodec {
    static Bytes encode(const std::vector<int>& values);
    static std::vector<int> decode(const Bytes& array);
};

std::string base64Encode(const Bytes& b);
Bytes base64Decode(const std::string& s);

// ---- the device library ---------------------------------------------------------------------------------
// libtsxform.so loaded by path (dlopen) so that tests can point the same host code at the emulated build.
class Backend {
public:
    explicit Backend(const std::string& libPath, int deviceIndex = 0);
    ~Backend();
    Backend(const Backend&) = delete;
    Backend& operator=(const Backend&) = delete;
    // one batch over host buffers; throws std::runtime_error on a batch-level failure, per-chunk status stays in descs
    void transformBatch(const tsx_batch_params& p, std::vector<tsx_chunk_desc>& descs, const uint8_t* src, size_t srcSize, uint8_t* dst, size_t dstSize);
    // the same with TSX_MEM_HOST_PACKED: transformed chunks back to back in dst, offsets and sizes returned in descs
    void transformBatchPacked(const tsx_batch_params& p, std::vector<tsx_chunk_desc>& descs, const uint8_t* src, size_t srcSize, uint8_t* dst, size_t dstSize);
    void detransformBatch(const tsx_batch_params& p, std::vector<tsx_chunk_desc>& descs, const uint8_t* src, size_t srcSize, uint8_t* dst, size_t dstSize);
    uint32_t crc32c(const uint8_t* data, size_t n);              // java.util.zip.CRC32C of one byte range (tsx_crc32c_batch)
    size_t transformedBound(size_t n, uint32_t flags) const;
    std::string strerror(int code) const;
    std::string version() const;

private:
    struct Fns;
    void* handle_ = nullptr;
    std::unique_ptr<Fns> f_;
    tsx_ctx* ctx_ = nullptr;        // this Backend's own context; a second concurrent caller borrows a pooled one (ctx == NULL in the ABI)
    struct Lock;
    std::unique_ptr<Lock> lock_;
};

// TransformedChunksSerializer / Deserializer: codec -> one Zstd frame (GPU compressor: same bytes as the reference's
// ZstdCompressCtx for this input) -> Base64, and back ("Invalid decompressed size: n" above 10 MiB).
std::string serializeTransformedChunks(Backend& be, const std::vector<int>& values, uint32_t zstdProfile = TSX_ZSTD_PROFILE_1_5_7);
std::vector<int> deserializeTransformedChunks(Backend& be, const std::string& base64);
// The chunk index as it appears in the segment manifest (Jackson layout of FixedSizeChunkIndex / VariableSizeChunkIndex,
// pinned by CT/manifest/index/ChunkIndexSerializationTest.java:63-123)
std::string chunkIndexToJson(Backend& be, const ChunkIndex& index);
std::shared_ptr<ChunkIndex> chunkIndexFromJson(Backend& be, const std::string& json);

// ---- streams (java.io.InputStream as far as the path needs it) --------------------------------------------
class InputStream {
public:
    virtual ~InputStream() = default;
    virtual long read(uint8_t* b, size_t len) = 0;   // InputStream.read(b, off, len): bytes read (> 0), -1 at end of stream
    virtual Bytes readNBytes(size_t n);               // up to n bytes, fewer only at end of stream
    virtual Bytes readAllBytes();
    virtual void close() {}
};
class ByteArrayInputStream : public InputStream {
public:
    explicit ByteArrayInputStream(Bytes data) : data_(std::move(data)) {}
    long read(uint8_t* b, size_t len) override;
    Bytes readNBytes(size_t n) override;
    Bytes readAllBytes() override;
    void close() override { closed_ = true; }
    bool closed() const { return closed_; }

private:
    Bytes data_;
    size_t pos_ = 0;
    bool closed_ = false;
};

// ---- security/DataKeyAndAAD ------------------------------------------------------------------------------
struct DataKeyAndAAD {
    Bytes dataKey;   // 32 bytes (AES-256, AesEncryptionProvider.java:36)
    Bytes aad;       // reference: 32 bytes
};
constexpr int IV_SIZE = 12;            // SegmentEncryptionMetadataV1.java:30
constexpr int GCM_TAG_BYTES = 16;      // AesEncryptionProvider.java:39 (128 bits)
// IV source; the default reads /dev/urandom (the reference: SecureRandom.getInstanceStrong(), AesEncryptionProvider.java:66-71)
using IvSupplier = std::function<void(uint8_t iv[IV_SIZE])>;
IvSupplier secureRandomIvSupplier();

// ---- upload side -------------------------------------------------------------------------------------------
class TransformChunkEnumeration {
public:
    virtual ~TransformChunkEnumeration() = default;
    virtual int originalChunkSize() const = 0;
    virtual std::optional<int> transformedChunkSize() const = 0;      // nullopt: unknown (variable)
    virtual bool hasMoreElements() = 0;
    virtual Bytes nextElement() = 0;                                  // throws std::out_of_range when exhausted
};

class BaseTransformChunkEnumeration : public TransformChunkEnumeration {
public:
    BaseTransformChunkEnumeration(std::shared_ptr<InputStream> inputStream, int originalChunkSize);   // 0 disables chunking
    int originalChunkSize() const override { return originalChunkSize_; }
    std::optional<int> transformedChunkSize() const override { return originalChunkSize_; }
    bool hasMoreElements() override;
    Bytes nextElement() override;

private:
    void fillChunkIfNeeded();
    std::shared_ptr<InputStream> in_;
    int originalChunkSize_;
    std::optional<Bytes> chunk_;
};

// Replaces CompressionChunkEnumeration + EncryptionChunkEnumeration (RemoteStorageManager.java:443-451) with one batched
// device call: reads ahead up to `batchChunks` chunks from `inner`, draws their IVs, transforms them together and hands
// them out in order.  transformedChunkSize(): unknown when compressing, else inner + 28 when encrypting.
class GpuTransformChunkEnumeration : public TransformChunkEnumeration {
public:
    // readAhead: while the consumer drains batch k (TransformFinisher hands its chunks to the uploader one by one), batch k + 1 is
    // already read from `inner` and on the device (one helper thread per enumeration; chunks, IVs and failures keep their order).
    // An upload thread then keeps two batches in flight instead of one and its uploads overlap the device - what keeps a GPU
    // fed by a handful of threads ("stragglers").  Off: `inner` is read exactly when the reference would read it.
    GpuTransformChunkEnumeration(std::shared_ptr<Backend> backend, std::shared_ptr<TransformChunkEnumeration> inner, bool compress,
                                 std::optional<DataKeyAndAAD> encryption, IvSupplier ivSupplier = secureRandomIvSupplier(),
                                 int batchChunks = 64, bool withCrc = false, uint32_t zstdProfile = TSX_ZSTD_PROFILE_1_5_7, bool readAhead = true);
    ~GpuTransformChunkEnumeration() override;
    int originalChunkSize() const override { return inner_->originalChunkSize(); }
    std::optional<int> transformedChunkSize() const override { return transformedChunkSize_; }
    bool hasMoreElements() override;
    Bytes nextElement() override;
    const std::vector<uint32_t>& crc32cOfOriginalChunks() const { return crcs_; }   // out of band, filled when withCrc
    // the next batch (up to batchChunks chunks of `inner`) transformed straight to the end of `object`, chunk
    // after chunk (TSX_MEM_HOST_PACKED) - the growing `.log` object or a multipart part buffer
    // (S3MultiPartOutputStream.java:89-122) - instead of one Bytes per chunk and a gather copy.  Appends each chunk's
    // transformed size to `sizes`; returns the bytes appended, 0 when `inner` is exhausted.  Do not mix with nextElement().
    size_t appendNextBatchPacked(Bytes& object, std::vector<int>& sizes);
    // ... and the batch on its own: the transformed chunks back to back, their sizes, their CRCs (GpuTransformFinisher reads it from here;
    // callable from a helper thread as long as nobody else touches this object meanwhile).  Empty sizes: `inner` is exhausted.
    struct PackedBatch { Bytes object; std::vector<int> sizes; std::vector<uint32_t> crcs; };
    PackedBatch transformNextBatchPacked();

private:
    struct Batch { std::vector<Bytes> chunks; std::vector<uint32_t> crcs; };
    Batch transformNextBatch();                        // pulls up to batch_ chunks from inner_; empty when inner_ is exhausted
    void fillBatchIfNeeded();
    std::shared_ptr<Backend> be_;
    std::shared_ptr<TransformChunkEnumeration> inner_;
    bool compress_;
    std::optional<DataKeyAndAAD> enc_;
    IvSupplier iv_;
    int batch_;
    bool withCrc_;
    uint32_t profile_;
    bool readAhead_;
    std::optional<int> transformedChunkSize_;
    std::vector<Bytes> ready_;
    size_t next_ = 0;
    std::vector<uint32_t> crcs_;
    std::future<Batch> ahead_;                         // the batch behind ready_, being transformed (readAhead_); the only toucher of inner_ while valid
    bool exhausted_ = false;
};

class TransformFinisher {
public:
    // chunkingEnabled == false: TransformFinisher.Builder.withChunkingDisabled()
    TransformFinisher(std::shared_ptr<TransformChunkEnumeration> inner, int originalFileSize, bool chunkingEnabled = true);
    bool hasMoreElements() { return inner_->hasMoreElements(); }
    Bytes nextElement();                                  // the reference wraps it in a ByteArrayInputStream
    std::shared_ptr<ChunkIndex> chunkIndex();             // "Chunk index was not built, was finisher used?"
    Bytes toBytes();                                      // SequenceInputStream(this) drained: the transformed .log object
    // the same object, assembled batch-wise in place when the chain is the GPU enumeration (no per-chunk Bytes, no gather copy)
    Bytes toBytesPacked();
    // TransformFinisher.toInputStream(): chunks are pulled (and transformed, batch-wise) only as the consumer reads;
    // with a bucket the stream is rate limited (TransformFinisher.java:146-151)
    std::shared_ptr<InputStream> toInputStream(std::shared_ptr<class TokenBucket> rateLimitingBucket = nullptr);

private:
    bool isBaseTransform() const;
    std::shared_ptr<TransformChunkEnumeration> inner_;
    std::unique_ptr<AbstractChunkIndexBuilder> builder_;
    int originalFileSize_;
    std::shared_ptr<ChunkIndex> chunkIndex_;
};

// SURVEY §8 f3, the upload sink without per-chunk arrays: TransformFinisher (TransformFinisher.java:48-151) for the GPU chain.  A batch of
// chunks is transformed back to back into ONE buffer (TSX_MEM_HOST_PACKED; on the JVM side a pinned direct ByteBuffer the compressor waves
// write into), the uploader's reads are served straight from it (SequenceInputStream semantics: a read never crosses a batch), the
// chunk index is fed from the batch's descriptor sizes, the rate limit wraps the stream as in the reference.  Twin of
// java/io/aiven/kafka/tieredstorage/gpu/GpuTransformFinisher.java - same members, same order of operations.
class GpuTransformFinisher {
public:
    GpuTransformFinisher(std::shared_ptr<GpuTransformChunkEnumeration> inner, int originalFileSize, bool chunkingEnabled = true,
                         std::shared_ptr<class TokenBucket> rateLimitingBucket = nullptr, bool readAhead = true);
    ~GpuTransformFinisher();
    std::shared_ptr<InputStream> toInputStream();          // TransformFinisher.toInputStream(): rate limited when a bucket was given
    // a sink with part buffers of its own (S3MultiPartOutputStream.java:89-122, partBuffer.put): fills `part` from the object, across
    // batches; returns the bytes written, less than `capacity` only at the object's end
    size_t fillPart(uint8_t* part, size_t capacity);
    size_t readSome(uint8_t* b, size_t len);               // at most what the current batch still holds; 0 at the object's end
    std::shared_ptr<ChunkIndex> chunkIndex();              // "Chunk index was not built, was finisher used?" until the object has been drained
    const std::vector<uint32_t>& crc32cOfOriginalChunks() const { return crcs_; }

private:
    bool nextBatch();
    std::shared_ptr<GpuTransformChunkEnumeration> inner_;
    std::unique_ptr<AbstractChunkIndexBuilder> builder_;
    std::shared_ptr<class TokenBucket> bucket_;
    bool readAhead_;
    Bytes cur_; size_t pos_ = 0;                           // the batch being read
    std::future<GpuTransformChunkEnumeration::PackedBatch> ahead_;     // the batch behind it, being transformed
    std::optional<int> pending_;                           // size of the newest chunk: addChunk or finish, once it is known which
    std::shared_ptr<ChunkIndex> chunkIndex_;
    std::vector<uint32_t> crcs_;
    bool exhausted_ = false;
};

// ---- fetch side --------------------------------------------------------------------------------------------
class DetransformChunkEnumeration {
public:
    virtual ~DetransformChunkEnumeration() = default;
    virtual bool hasMoreElements() = 0;
    virtual Bytes nextElement() = 0;
};

class BaseDetransformChunkEnumeration : public DetransformChunkEnumeration {
public:
    explicit BaseDetransformChunkEnumeration(std::shared_ptr<InputStream> inputStream);                       // no chunking: everything at once
    BaseDetransformChunkEnumeration(std::shared_ptr<InputStream> inputStream, std::vector<Chunk> chunks);
    bool hasMoreElements() override;
    Bytes nextElement() override;
    std::shared_ptr<InputStream> inputStream() const { return in_; }

private:
    void fillChunkIfNeeded();
    std::shared_ptr<InputStream> in_;
    bool inputStreamClosed_ = false;
    std::vector<Chunk> chunks_;
    size_t iter_ = 0;
    bool isEmpty_;
    std::optional<Bytes> chunk_;
};

struct SegmentEncryptionMetadata { Bytes dataKey; Bytes aad; int ivSize = IV_SIZE; };

// Replaces DecryptionChunkEnumeration + DecompressionChunkEnumeration (DefaultChunkManager.java:58-67).
// Failures surface as the reference's do: std::runtime_error("Tag mismatch") (javax.crypto.AEADBadTagException wrapped),
// std::runtime_error("Invalid decompressed size: n"), std::runtime_error(corrupt frame).
class GpuDetransformChunkEnumeration : public DetransformChunkEnumeration {
public:
    GpuDetransformChunkEnumeration(std::shared_ptr<Backend> backend, std::shared_ptr<DetransformChunkEnumeration> inner, bool compressed,
                                   std::optional<SegmentEncryptionMetadata> encryption, int maxOriginalChunkSize, int batchChunks = 64);
    bool hasMoreElements() override;
    Bytes nextElement() override;

private:
    void fillBatchIfNeeded();
    std::shared_ptr<Backend> be_;
    std::shared_ptr<DetransformChunkEnumeration> inner_;
    bool compressed_;
    std::optional<SegmentEncryptionMetadata> enc_;
    int maxOriginal_, batch_;
    std::vector<Bytes> ready_;
    size_t next_ = 0;
    std::optional<std::string> failMsg_;      // failure of the chunk that follows ready_
};

class DetransformFinisher {
public:
    explicit DetransformFinisher(std::shared_ptr<DetransformChunkEnumeration> inner) : inner_(std::move(inner)) {}
    Bytes toBytes();      // toInputStream() drained; a pure base enumeration hands back the raw stream's bytes

private:
    std::shared_ptr<DetransformChunkEnumeration> inner_;
};

// storage/ObjectFetcher.java:27-35 as far as the path needs it
class ObjectFetcher {
public:
    virtual ~ObjectFetcher() = default;
    virtual std::shared_ptr<InputStream> fetch(const std::string& objectKey, BytesRange range) = 0;
};

struct SegmentManifest {                 // manifest/SegmentManifest.java:36-46 as far as the fetch path needs it
    std::shared_ptr<ChunkIndex> chunkIndex;
    bool compression = false;
    std::optional<SegmentEncryptionMetadata> encryption;
};

// manifest/SegmentIndexV1.java:26-48, SegmentIndexesV1.java:26-82, SegmentManifestV1.java:30-132: the manifest object as it is
// stored next to the .log object - the JSON a reference broker reads back (Jackson layout pinned by
// CT/manifest/SegmentManifestV1SerdeTest.java:82-133).  Two pieces stay opaque strings because their producers stay Java: the
// RSA-wrapped data key "<keyId>:<base64>" (EncryptedDataKey.serialize, DataKeySerializer.java:30-46) and Kafka's
// RemoteLogSegmentMetadata (KafkaTypeSerdeModule; written, never read back: JsonProperty.Access.READ_ONLY).
struct SegmentIndexV1 { int position = 0, size = 0; bool operator==(const SegmentIndexV1& o) const { return position == o.position && size == o.size; } };
struct SegmentIndexesV1 {
    SegmentIndexV1 offset, timestamp, producerSnapshot, leaderEpoch;
    std::optional<SegmentIndexV1> transaction;                        // "transaction":null when the segment has no txn index
};
// manifest/SegmentIndexesV1Builder.java:27-66: the index files of a segment follow one another in the `.indexes` object (each one ONE
// encrypt-only chunk, transformIndex below); add() records (running position, transformed size) per index type.
enum class IndexType { OFFSET = 0, TIMESTAMP = 1, PRODUCER_SNAPSHOT = 2, TRANSACTION = 3, LEADER_EPOCH = 4 };   // Kafka's RemoteStorageManager.IndexType order
class SegmentIndexesV1Builder {
public:
    SegmentIndexesV1Builder& add(IndexType type, int size);       // std::logic_error("Index OFFSET is already added")
    std::vector<IndexType> indexes() const;                        // sorted, for messages
    SegmentIndexesV1 build() const;                                // the reference's two IllegalStateException messages
private:
    std::vector<std::pair<IndexType, SegmentIndexV1>> added_;
    int currentPosition_ = 0;
};

struct SegmentManifestV1 : SegmentManifest {
    SegmentIndexesV1 segmentIndexes;
    std::string remoteLogSegmentMetadataJson;                         // "" = property absent
};
using DataKeyEncryptor = std::function<std::string(const Bytes& dataKey)>;      // -> "<keyId>:<base64 of the RSA-wrapped key>"
using DataKeyDecryptor = std::function<Bytes(const std::string& serialized)>;
std::string segmentManifestToJson(Backend& be, const SegmentManifestV1& m, const DataKeyEncryptor& wrapKey = nullptr);
SegmentManifestV1 segmentManifestFromJson(Backend& be, const std::string& json, const DataKeyDecryptor& unwrapKey = nullptr);

class ChunkManager {
public:
    virtual ~ChunkManager() = default;
    virtual Bytes getChunk(const std::string& objectKey, const SegmentManifest& manifest, int chunkId) = 0;
};

class GpuChunkManager : public ChunkManager {
public:
    GpuChunkManager(std::shared_ptr<Backend> backend, std::shared_ptr<ObjectFetcher> fetcher) : be_(std::move(backend)), fetcher_(std::move(fetcher)) {}
    Bytes getChunk(const std::string& objectKey, const SegmentManifest& manifest, int chunkId) override;
    // a prefetch window of consecutive chunks in ONE ranged fetch and ONE device batch
    std::vector<Bytes> getChunks(const std::string& objectKey, const SegmentManifest& manifest, int firstChunkId, int count);

private:
    std::shared_ptr<Backend> be_;
    std::shared_ptr<ObjectFetcher> fetcher_;
};

// ---- fetch-side batching -------------------------------------------------------------------
// ChunkCache.java:76-129 (getChunk) + 159-184 (startPrefetching) with the device in mind.  The reference turns a prefetch window
// of k chunks into k single-chunk tasks, each one ranged fetch + one detransform; through a GPU that is k launches of a kernel
// whose latency is per chunk.  Here the requested chunk and the not-yet-cached part of its prefetch window become ONE
// GpuChunkManager::getChunks call (one ranged fetch, one device batch), and concurrent misses on adjacent chunks of the same object
// that arrive within a short bounded wait (coalesceWait, far below get.timeout.ms) join the batch that is about to leave.  Kept from
// the reference: nothing beyond the configured window is ever fetched (laziness), a cached chunk is returned as a
// copy of its bytes (eviction cannot pull it from under a reader), a waiter gives up after getTimeout with a RuntimeException, and a
// chunk that fails (tag mismatch, corrupt frame) fails only the callers that asked for that chunk - the batch is retried chunk by chunk.
struct ChunkCacheStats { long hits = 0, misses = 0, fetchCalls = 0, chunksFetched = 0, joined = 0, evictions = 0; };
class GpuChunkCache : public ChunkManager {
public:
    GpuChunkCache(std::shared_ptr<GpuChunkManager> manager, int prefetchingSize, size_t maxBytes, int getTimeoutMs = 10000, int coalesceWaitMicros = 300);
    ~GpuChunkCache() override;
    Bytes getChunk(const std::string& objectKey, const SegmentManifest& manifest, int chunkId) override;
    ChunkCacheStats stats() const;
    void quiesce();                    // waits for the prefetch batches nobody is waiting for (tests; orderly shutdown)

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// fetch/FetchChunkEnumeration.java:40-178: the caller of ChunkManager.getChunk on fetchLogSegment().  An original-offset range
// [from, to] (inclusive; to beyond the end is clamped to the last chunk) becomes the chunks that cover it, the first one skipped to
// `from`, the last one bounded at `to`.  Lazy: a chunk is asked of the ChunkManager only when nextElement() reaches it, and after
// close() nothing more is asked for - with a GpuChunkCache underneath that means: nothing beyond the cache's configured window.
class FetchChunkEnumeration {
public:
    FetchChunkEnumeration(std::shared_ptr<ChunkManager> chunkManager, std::string objectKey, SegmentManifest manifest, BytesRange range);
    bool hasMoreElements() const { return !closed_ && currentChunkId_ <= lastChunkId_; }
    Bytes nextElement();                      // the part of the next chunk that lies inside the range; std::out_of_range (NoSuchElementException) at the end
    Bytes readAll();                          // toInputStream().readAllBytes()
    void close() { closed_ = true; }
    int startChunkId() const { return startChunkId_; }
    int lastChunkId() const { return lastChunkId_; }
    int currentChunkId() const { return currentChunkId_; }

private:
    std::shared_ptr<ChunkManager> chunkManager_;
    std::string objectKey_;
    SegmentManifest manifest_;
    BytesRange range_;
    int startChunkId_ = 0, lastChunkId_ = 0, currentChunkId_ = 0;
    bool closed_ = false;
};

// ---- upload sink --------------------------------------------------------------------------
// io.github.bucket4j.Bucket as RateLimitedInputStream.rateLimitBucket builds it (RateLimitedInputStream.java:46-55):
// capacity = rate tokens, greedy refill of `rate` tokens per second, starts full.
class TokenBucket {
public:
    explicit TokenBucket(int uploadRate);                 // rate = max(uploadRate, MIN_RATE)
    void consume(long tokens);                            // asBlocking().consume: sleeps until the tokens are there
    void forceAddTokens(long tokens);
    static constexpr int MIN_RATE = 8192;                 // InputStream.DEFAULT_BUFFER_SIZE before JDK 21

private:
    void refill();
    double tokens_, rate_;
    long long lastNs_;
};
class RateLimitedInputStream : public InputStream {      // RateLimitedInputStream.java:56-84: only read(b, off, len) is limited
public:
    RateLimitedInputStream(std::shared_ptr<InputStream> delegated, std::shared_ptr<TokenBucket> bucket) : in_(std::move(delegated)), bucket_(std::move(bucket)) {}
    long read(uint8_t* b, size_t len) override;
    void close() override { in_->close(); }

private:
    std::shared_ptr<InputStream> in_;
    std::shared_ptr<TokenBucket> bucket_;
};
class ObjectUploader {                                   // storage/ObjectUploader.java:27
public:
    virtual ~ObjectUploader() = default;
    virtual long upload(InputStream& inputStream, const std::string& objectKey) = 0;     // returns the object's size
};
// storage/filesystem/FileSystemStorage.java:41-90: objects are files under a root directory
class FileSystemStorage : public ObjectUploader, public ObjectFetcher {
public:
    explicit FileSystemStorage(std::string root);        // "<root> must be a writable directory"
    long upload(InputStream& inputStream, const std::string& objectKey) override;
    std::shared_ptr<InputStream> fetch(const std::string& objectKey, BytesRange range) override;

private:
    std::string root_;
};

// ---- neighbours of the path ------------------------------------------------------------------
// SegmentCompressionChecker.check (core/.../SegmentCompressionChecker.java:37-53): is the segment's first record batch
// compressed?  Kafka's RecordBatch.ensureValid() for magic v2 = size sanity + CRC32C over [attributes .. end of batch];
// the CRC runs on the device.  Throws InvalidRecordBatchException (std::runtime_error) like the reference.
struct InvalidRecordBatchException : std::runtime_error { using std::runtime_error::runtime_error; };
bool segmentIsCompressed(Backend& be, const Bytes& segmentHead);

// RemoteStorageManager.transformIndex (core/.../RemoteStorageManager.java:455-490): an index file is ONE chunk, never
// compressed, encrypted when encryption is on; returns the transformed bytes (size 0 -> empty, recorded as size 0).
Bytes transformIndex(std::shared_ptr<Backend> be, const Bytes& index, const std::optional<DataKeyAndAAD>& key, IvSupplier iv = secureRandomIvSupplier());

}  // namespace tsx
