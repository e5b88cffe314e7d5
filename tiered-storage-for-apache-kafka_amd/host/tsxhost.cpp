// tsxhost — see tsxhost.hpp.  Host logic only: every byte of compression / encryption / checksum is computed by
// libtsxform.so on the GPU (there is no CPU fallback here either: a missing library or device throws).
#include "tsxhost.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <future>
#include <list>
#include <map>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>

namespace tsx {

// =====================================================================================================
// chunk index (manifest/index/*.java)
// =====================================================================================================
static void checkSizeNonNegative(int size, const char* name) {
    if (size < 0) throw std::invalid_argument(std::string(name) + " must be non-negative, " + std::to_string(size) + " given");
}
static void checkSizePositive(int size, const char* name) {
    if (size <= 0) throw std::invalid_argument(std::string(name) + " must be positive, " + std::to_string(size) + " given");
}

ChunkIndex::ChunkIndex(int originalChunkSize, int originalFileSize, int finalTransformedChunkSize, int chunkCount)
    : originalChunkSize_(originalChunkSize), originalFileSize_(originalFileSize), finalTransformedChunkSize_(finalTransformedChunkSize),
      chunkCount_(chunkCount) {
    checkSizePositive(originalChunkSize, "Original chunk size");
    checkSizeNonNegative(originalFileSize, "Original file size");
    checkSizeNonNegative(finalTransformedChunkSize, "Final transformed chunk size");
}

int ChunkIndex::originalChunkSizeOf(int chunkI) const {
    const bool isFinalChunk = chunkI == chunkCount_ - 1;
    return isFinalChunk ? (originalFileSize_ - (chunkCount_ - 1) * originalChunkSize_) : originalChunkSize_;
}

void ChunkIndex::materializeChunks() {                       // AbstractChunkIndex.java:52-72
    chunks_.clear();
    int originalPosition = 0, transformedPosition = 0;
    if (chunkCount_ == 0) {
        chunks_.push_back(Chunk{0, 0, 0, 0, 0});
        return;
    }
    for (int chunkI = 0; chunkI < chunkCount_; chunkI++) {
        const int originalSize = originalChunkSizeOf(chunkI), transformedSize = transformedChunkSize(chunkI);
        chunks_.push_back(Chunk{chunkI, originalPosition, originalSize, transformedPosition, transformedSize});
        originalPosition += originalSize;
        transformedPosition += transformedSize;
    }
}

std::optional<Chunk> ChunkIndex::findChunkForOriginalOffset(int offset) const {      // AbstractChunkIndex.java:75-110
    if (offset < 0) throw std::invalid_argument("Offset must be non-negative, " + std::to_string(offset) + " given");
    if (offset >= originalFileSize_) return std::nullopt;
    int chunkI = 0, curOriginalChunkPosition = 0, curTransformedChunkPosition = 0;
    for (; chunkI < chunkCount_; chunkI++) {
        const int firstOffsetBeyondCurOriginalChunk = (chunkI + 1) * originalChunkSize_;
        if (offset < firstOffsetBeyondCurOriginalChunk) break;
        curOriginalChunkPosition += originalChunkSizeOf(chunkI);
        curTransformedChunkPosition += transformedChunkSize(chunkI);
    }
    return Chunk{chunkI, curOriginalChunkPosition, originalChunkSizeOf(chunkI), curTransformedChunkPosition, transformedChunkSize(chunkI)};
}

std::vector<Chunk> ChunkIndex::chunksForRange(BytesRange r) const {                  // AbstractChunkIndex.java:113-123
    std::vector<Chunk> result;
    for (int i = r.from; i <= r.to && i < originalFileSize_;) {
        const Chunk c = *findChunkForOriginalOffset(i);
        result.push_back(c);
        i += c.originalSize;
    }
    return result;
}

static int fixedChunkCount(int originalChunkSize, int originalFileSize) {
    checkSizePositive(originalChunkSize, "Original chunk size");
    return originalFileSize % originalChunkSize == 0 ? originalFileSize / originalChunkSize : originalFileSize / originalChunkSize + 1;
}

FixedSizeChunkIndex::FixedSizeChunkIndex(int originalChunkSize, int originalFileSize, int transformedChunkSize, int finalTransformedChunkSize)
    : ChunkIndex(originalChunkSize, originalFileSize, finalTransformedChunkSize, fixedChunkCount(originalChunkSize, originalFileSize)),
      transformedChunkSize_(transformedChunkSize) {
    checkSizeNonNegative(transformedChunkSize, "Transformed chunk size");
    materializeChunks();
}

VariableSizeChunkIndex::VariableSizeChunkIndex(int originalChunkSize, int originalFileSize, std::vector<int> transformedChunks)
    : ChunkIndex(originalChunkSize, originalFileSize, transformedChunks.empty() ? 0 : transformedChunks.back(), (int)transformedChunks.size()),
      transformedChunks_(std::move(transformedChunks)) {
    materializeChunks();
}

AbstractChunkIndexBuilder::AbstractChunkIndexBuilder(int originalChunkSize, int originalFileSize) {
    checkSize(originalChunkSize, "Original chunk size");
    originalChunkSize_ = originalChunkSize;
    checkSize(originalFileSize, "Original file size");
    originalFileSize_ = originalFileSize;
}
void AbstractChunkIndexBuilder::checkSize(int size, const char* name) { checkSizeNonNegative(size, name); }

void AbstractChunkIndexBuilder::addChunk(int transformedChunkSize) {                 // AbstractChunkIndexBuilder.java:39-55
    if (finished_) throw std::logic_error("Cannot add chunk to already finished index");
    checkSize(transformedChunkSize, "Transformed chunk size");
    if (remainOfOriginalFileSize() <= originalChunkSize_) throw std::logic_error("This must be final chunk. Call `finish` instead.");
    addChunk0(transformedChunkSize);
    chunksAdded_ += 1;
}

std::shared_ptr<ChunkIndex> AbstractChunkIndexBuilder::finish(int finalTransformedChunkSize) {     // :63-83
    if (finished_) throw std::logic_error("Cannot finish already finished index");
    checkSize(finalTransformedChunkSize, "Transformed chunk size");
    if (remainOfOriginalFileSize() > originalChunkSize_)
        throw std::logic_error("This cannot be final chunk: not enough chunks to cover original file. Call `addChunk` instead.");
    auto result = finish0(finalTransformedChunkSize);
    chunksAdded_ += 1;
    finished_ = true;
    return result;
}

FixedSizeChunkIndexBuilder::FixedSizeChunkIndexBuilder(int originalChunkSize, int originalFileSize, int transformedChunkSize)
    : AbstractChunkIndexBuilder(originalChunkSize, originalFileSize) {
    checkSize(transformedChunkSize, "Transformed chunk size");
    transformedChunkSize_ = transformedChunkSize;
}
void FixedSizeChunkIndexBuilder::addChunk0(int transformedChunkSize) {               // FixedSizeChunkIndexBuilder.java:31-38
    if (transformedChunkSize != transformedChunkSize_)
        throw std::invalid_argument("Non-final chunk must be of size " + std::to_string(transformedChunkSize_) + ", but " +
                                    std::to_string(transformedChunkSize) + " given");
}
std::shared_ptr<ChunkIndex> FixedSizeChunkIndexBuilder::finish0(int finalTransformedChunkSize) {
    return std::make_shared<FixedSizeChunkIndex>(originalChunkSize_, originalFileSize_, transformedChunkSize_, finalTransformedChunkSize);
}
std::shared_ptr<ChunkIndex> VariableSizeChunkIndexBuilder::finish0(int finalTransformedChunkSize) {
    transformedChunks_.push_back(finalTransformedChunkSize);
    return std::make_shared<VariableSizeChunkIndex>(originalChunkSize_, originalFileSize_, transformedChunks_);
}

// ---- ChunkSizesBinaryCodec.java:104-202 (big-endian) ------------------------------------------------------
static void putInt(Bytes& b, uint32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
static uint32_t getInt(const Bytes& b, size_t& pos) {
    if (pos + 4 > b.size()) throw std::runtime_error("BufferUnderflowException");
    uint32_t v = ((uint32_t)b[pos] << 24) | ((uint32_t)b[pos + 1] << 16) | ((uint32_t)b[pos + 2] << 8) | b[pos + 3];
    pos += 4;
    return v;
}
static int bytesNeeded(int v) { return v <= 0xFF ? 1 : v <= 0xFFFF ? 2 : v <= 0xFFFFFF ? 3 : 4; }

Bytes ChunkSizesBinaryCodec::encode(const std::vector<int>& values) {
    Bytes out;
    const int count = (int)values.size();
    putInt(out, (uint32_t)count);
    if (count == 0) return out;
    const int lastValue = values.back();
    if (count == 1) {
        if (lastValue < 0) throw std::invalid_argument("Values cannot be negative");
        putInt(out, (uint32_t)lastValue);
        return out;
    }
    const int min = *std::min_element(values.begin(), values.end() - 1);
    if (min < 0 || lastValue < 0) throw std::invalid_argument("Values cannot be negative");
    const int base = min;
    int bytesPerValue = 0;
    for (int i = 0; i < count - 1; i++) bytesPerValue = std::max(bytesPerValue, bytesNeeded(values[(size_t)i] - base));
    putInt(out, (uint32_t)base);
    out.push_back((uint8_t)bytesPerValue);
    for (int i = 0; i < count - 1; i++) {
        const uint32_t onBase = (uint32_t)(values[(size_t)i] - base);
        for (int s = 8 * (bytesPerValue - 1); s >= 0; s -= 8) out.push_back((uint8_t)(onBase >> s));
    }
    putInt(out, (uint32_t)lastValue);
    return out;
}

std::vector<int> ChunkSizesBinaryCodec::decode(const Bytes& a) {
    size_t pos = 0;
    const int count = (int)getInt(a, pos);
    if (count == 0) return {};
    if (count == 1) return {(int)getInt(a, pos)};
    std::vector<int> result;
    const int base = (int)getInt(a, pos);
    if (pos >= a.size()) throw std::runtime_error("BufferUnderflowException");
    const int bytesPerValue = a[pos++];
    for (int i = 0; i < count - 1; i++) {
        if (pos + (size_t)bytesPerValue > a.size()) throw std::runtime_error("BufferUnderflowException");
        uint32_t v = 0;
        for (int k = 0; k < bytesPerValue; k++) v = (v << 8) | a[pos++];
        result.push_back((int)v + base);
    }
    result.push_back((int)getInt(a, pos));
    return result;
}

static const char kB64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
std::string base64Encode(const Bytes& b) {
    std::string s;
    size_t i = 0;
    for (; i + 2 < b.size(); i += 3) {
        const uint32_t v = ((uint32_t)b[i] << 16) | ((uint32_t)b[i + 1] << 8) | b[i + 2];
        s += kB64[v >> 18]; s += kB64[(v >> 12) & 63]; s += kB64[(v >> 6) & 63]; s += kB64[v & 63];
    }
    if (i + 1 == b.size()) { const uint32_t v = (uint32_t)b[i] << 16; s += kB64[v >> 18]; s += kB64[(v >> 12) & 63]; s += "=="; }
    else if (i + 2 == b.size()) { const uint32_t v = ((uint32_t)b[i] << 16) | ((uint32_t)b[i + 1] << 8); s += kB64[v >> 18]; s += kB64[(v >> 12) & 63]; s += kB64[(v >> 6) & 63]; s += '='; }
    return s;
}
Bytes base64Decode(const std::string& s) {
    Bytes out; uint32_t acc = 0; int bits = 0;
    for (char c : s) {
        if (c == '=') break;
        const char* p = strchr(kB64, c);
        if (!p || !c) throw std::invalid_argument("Illegal base64 character");
        acc = (acc << 6) | (uint32_t)(p - kB64); bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return out;
}

// =====================================================================================================
// Backend: libtsxform.so through its C ABI
// =====================================================================================================
struct Backend::Fns {
    decltype(&tsx_init) init; decltype(&tsx_ctx_create) ctx_create; decltype(&tsx_ctx_destroy) ctx_destroy;
    decltype(&tsx_transform_batch) transform; decltype(&tsx_detransform_batch) detransform;
    decltype(&tsx_crc32c_batch) crc; decltype(&tsx_transformed_bound) bound; decltype(&tsx_strerror) strerr; decltype(&tsx_version) version; decltype(&tsx_abi_version) abi;
};

struct Backend::Lock { std::mutex mu; };
namespace {
struct CtxLease {                    // the Backend's own context if it is free, else NULL = a pooled context of the library
    CtxLease(std::mutex& m, tsx_ctx* own) : mu(m), got(m.try_lock()), ctx(got ? own : nullptr) {}
    ~CtxLease() { if (got) mu.unlock(); }
    std::mutex& mu; bool got; tsx_ctx* ctx;
};
}  // namespace

Backend::Backend(const std::string& libPath, int deviceIndex) : f_(new Fns), lock_(new Lock) {
    handle_ = dlopen(libPath.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!handle_) throw std::runtime_error("tsxhost: cannot load " + libPath + ": " + dlerror() + " (there is no CPU fallback)");
    auto sym = [&](const char* n) { void* p = dlsym(handle_, n); if (!p) throw std::runtime_error(std::string("tsxhost: missing symbol ") + n); return p; };
    f_->init = (decltype(f_->init))sym("tsx_init"); f_->ctx_create = (decltype(f_->ctx_create))sym("tsx_ctx_create");
    f_->ctx_destroy = (decltype(f_->ctx_destroy))sym("tsx_ctx_destroy"); f_->transform = (decltype(f_->transform))sym("tsx_transform_batch");
    f_->detransform = (decltype(f_->detransform))sym("tsx_detransform_batch"); f_->bound = (decltype(f_->bound))sym("tsx_transformed_bound");
    f_->crc = (decltype(f_->crc))sym("tsx_crc32c_batch");
    f_->strerr = (decltype(f_->strerr))sym("tsx_strerror"); f_->version = (decltype(f_->version))sym("tsx_version");
    f_->abi = (decltype(f_->abi))sym("tsx_abi_version");
    if (f_->abi() != TSX_ABI_VERSION) throw std::runtime_error("tsxhost: ABI version mismatch");
    const int n = f_->init(0, nullptr);
    if (n <= 0) throw std::runtime_error(std::string("tsxhost: tsx_init failed: ") + f_->strerr(n));
    const int rc = f_->ctx_create(deviceIndex, 0, 0, &ctx_);
    if (rc) throw std::runtime_error(std::string("tsxhost: tsx_ctx_create failed: ") + f_->strerr(rc));
}
Backend::~Backend() {
    if (ctx_) f_->ctx_destroy(ctx_);
    // the library stays loaded: other Backends of the process share its device state
}
void Backend::transformBatch(const tsx_batch_params& p, std::vector<tsx_chunk_desc>& d, const uint8_t* src, size_t srcSize, uint8_t* dst, size_t dstSize) {
    CtxLease lease(lock_->mu, ctx_);
    const int rc = f_->transform(lease.ctx, &p, d.data(), (uint32_t)d.size(), src, srcSize, dst, dstSize, TSX_MEM_HOST);
    if (rc) throw std::runtime_error(std::string("tsx_transform_batch: ") + f_->strerr(rc));
}
void Backend::transformBatchPacked(const tsx_batch_params& p, std::vector<tsx_chunk_desc>& d, const uint8_t* src, size_t srcSize, uint8_t* dst, size_t dstSize) {
    CtxLease lease(lock_->mu, ctx_);
    const int rc = f_->transform(lease.ctx, &p, d.data(), (uint32_t)d.size(), src, srcSize, dst, dstSize, TSX_MEM_HOST_PACKED);
    if (rc) throw std::runtime_error(std::string("tsx_transform_batch: ") + f_->strerr(rc));
}
void Backend::detransformBatch(const tsx_batch_params& p, std::vector<tsx_chunk_desc>& d, const uint8_t* src, size_t srcSize, uint8_t* dst, size_t dstSize) {
    CtxLease lease(lock_->mu, ctx_);
    const int rc = f_->detransform(lease.ctx, &p, d.data(), (uint32_t)d.size(), src, srcSize, dst, dstSize, TSX_MEM_HOST);
    if (rc) throw std::runtime_error(std::string("tsx_detransform_batch: ") + f_->strerr(rc));
}
uint32_t Backend::crc32c(const uint8_t* data, size_t n) {
    Bytes buf(((n + 15) & ~(size_t)15) + 16);                       // the ABI wants a 16-byte aligned chunk start
    if (n) memcpy(buf.data(), data, n);
    tsx_chunk_desc d;
    memset(&d, 0, sizeof d);
    d.src_len = (uint32_t)n;
    CtxLease lease(lock_->mu, ctx_);
    const int rc = f_->crc(lease.ctx, &d, 1, buf.data(), buf.size(), TSX_MEM_HOST);
    if (rc || d.status != TSX_OK) throw std::runtime_error(std::string("tsx_crc32c_batch: ") + f_->strerr(rc ? rc : d.status));
    return d.crc32c;
}
size_t Backend::transformedBound(size_t n, uint32_t flags) const { return f_->bound(n, flags); }
std::string Backend::strerror(int code) const { return f_->strerr(code); }
std::string Backend::version() const { return f_->version(); }

static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

static tsx_batch_params makeParams(uint32_t flags, const Bytes* key, const Bytes* aad, uint32_t profile) {
    tsx_batch_params p;
    memset(&p, 0, sizeof p);
    p.flags = flags;
    p.zstd_level = 0;
    p.zstd_profile = profile;
    if (flags & TSX_ENCRYPT) {
        if (!key || key->size() != 32) throw std::invalid_argument("AES-256 data key must be 32 bytes");
        if (aad && aad->size() > sizeof p.aad) throw std::invalid_argument("AAD longer than 64 bytes");
        memcpy(p.key, key->data(), 32);
        if (aad) { memcpy(p.aad, aad->data(), aad->size()); p.aad_len = (uint32_t)aad->size(); }
    }
    return p;
}

// Zstd frame header (RFC 8878 3.1.1.1): content size, -1 unknown / not a frame  (Zstd.decompressedSize)
static long long zstdFrameContentSize(const Bytes& f) {
    if (f.size() < 5 || f[0] != 0x28 || f[1] != 0xB5 || f[2] != 0x2F || f[3] != 0xFD) return -1;
    const uint8_t fhd = f[4];
    const int fcsFlag = fhd >> 6, single = (fhd >> 5) & 1, dictFlag = fhd & 3;
    size_t pos = 5 + (single ? 0 : 1) + (dictFlag == 0 ? 0 : dictFlag == 1 ? 1 : dictFlag == 2 ? 2 : 4);
    const int fcsBytes = fcsFlag == 0 ? (single ? 1 : 0) : fcsFlag == 1 ? 2 : fcsFlag == 2 ? 4 : 8;
    if (fcsBytes == 0 || pos + (size_t)fcsBytes > f.size()) return -1;
    unsigned long long v = 0;
    for (int i = 0; i < fcsBytes; i++) v |= (unsigned long long)f[pos + (size_t)i] << (8 * i);
    if (fcsBytes == 2) v += 256;
    return (long long)v;
}

std::string serializeTransformedChunks(Backend& be, const std::vector<int>& values, uint32_t profile) {
    const Bytes bin = ChunkSizesBinaryCodec::encode(values);
    if (bin.size() > 10u * 1024 * 1024) throw std::invalid_argument("Encoded index is too big (" + std::to_string(bin.size()) + "), cannot serialize");
    Bytes src(align16(bin.size()) + 16);
    memcpy(src.data(), bin.data(), bin.size());
    const size_t cap = be.transformedBound(bin.size(), TSX_COMPRESS);
    Bytes dst(align16(cap) + 16);
    std::vector<tsx_chunk_desc> d(1);
    memset(d.data(), 0, sizeof(tsx_chunk_desc));
    d[0].src_len = (uint32_t)bin.size(); d[0].dst_cap = (uint32_t)cap;
    const tsx_batch_params p = makeParams(TSX_COMPRESS, nullptr, nullptr, profile);
    be.transformBatch(p, d, src.data(), src.size(), dst.data(), dst.size());
    if (d[0].status != TSX_OK) throw std::runtime_error("index compression failed: " + be.strerror(d[0].status));
    dst.resize(d[0].dst_len);
    return base64Encode(dst);
}

std::vector<int> deserializeTransformedChunks(Backend& be, const std::string& base64) {
    const Bytes frame = base64Decode(base64);
    const long long size = zstdFrameContentSize(frame);
    if (size < 0 || size > 10ll * 1024 * 1024) throw std::runtime_error("Invalid decompressed size: " + std::to_string(size));
    Bytes src(align16(frame.size()) + 16);
    memcpy(src.data(), frame.data(), frame.size());
    Bytes dst(align16((size_t)size) + 16);
    std::vector<tsx_chunk_desc> d(1);
    memset(d.data(), 0, sizeof(tsx_chunk_desc));
    d[0].src_len = (uint32_t)frame.size(); d[0].dst_cap = (uint32_t)size;
    const tsx_batch_params p = makeParams(TSX_COMPRESS, nullptr, nullptr, TSX_ZSTD_PROFILE_1_5_7);
    be.detransformBatch(p, d, src.data(), src.size(), dst.data(), dst.size());
    if (d[0].status != TSX_OK) throw std::runtime_error("index decompression failed: " + be.strerror(d[0].status));
    dst.resize(d[0].dst_len);
    return ChunkSizesBinaryCodec::decode(dst);
}

std::string chunkIndexToJson(Backend& be, const ChunkIndex& index) {
    std::string j = "{\"type\":\"";
    j += index.isFixed() ? "fixed" : "variable";
    j += "\",\"originalChunkSize\":" + std::to_string(index.originalChunkSize()) + ",\"originalFileSize\":" + std::to_string(index.originalFileSize());
    if (index.isFixed()) {
        const auto& f = static_cast<const FixedSizeChunkIndex&>(index);
        j += ",\"transformedChunkSize\":" + std::to_string(f.transformedChunkSize()) + ",\"finalTransformedChunkSize\":" + std::to_string(f.finalTransformedChunkSize());
    } else {
        j += ",\"transformedChunks\":\"" + serializeTransformedChunks(be, static_cast<const VariableSizeChunkIndex&>(index).transformedChunks()) + "\"";
    }
    return j + "}";
}

static std::string jsonField(const std::string& json, const std::string& name) {      // flat object, no escapes in the values we read
    const std::string key = "\"" + name + "\":";
    const size_t k = json.find(key);
    if (k == std::string::npos) throw std::invalid_argument("Missing required creator property '" + name + "'");
    size_t b = k + key.size(), e;
    if (json[b] == '"') { b++; e = json.find('"', b); }
    else e = json.find_first_of(",}", b);
    if (e == std::string::npos) throw std::invalid_argument("malformed JSON");
    return json.substr(b, e - b);
}

std::shared_ptr<ChunkIndex> chunkIndexFromJson(Backend& be, const std::string& json) {
    const std::string type = jsonField(json, "type");
    const int ocs = std::stoi(jsonField(json, "originalChunkSize")), ofs = std::stoi(jsonField(json, "originalFileSize"));
    if (type == "fixed")
        return std::make_shared<FixedSizeChunkIndex>(ocs, ofs, std::stoi(jsonField(json, "transformedChunkSize")), std::stoi(jsonField(json, "finalTransformedChunkSize")));
    if (type == "variable") return std::make_shared<VariableSizeChunkIndex>(ocs, ofs, deserializeTransformedChunks(be, jsonField(json, "transformedChunks")));
    throw std::invalid_argument("Could not resolve type id '" + type + "'");
}

// =====================================================================================================
// streams, IVs
// =====================================================================================================
Bytes InputStream::readNBytes(size_t n) {                              // InputStream.readNBytes: loops over read() until n or end
    Bytes out(n);
    size_t got = 0;
    while (got < n) {
        const long r = read(out.data() + got, std::min<size_t>(n - got, 8192));
        if (r <= 0) break;
        got += (size_t)r;
    }
    out.resize(got);
    return out;
}
Bytes InputStream::readAllBytes() {
    Bytes out;
    uint8_t buf[8192];
    for (;;) { const long r = read(buf, sizeof buf); if (r <= 0) break; out.insert(out.end(), buf, buf + r); }
    return out;
}
long ByteArrayInputStream::read(uint8_t* b, size_t len) {
    if (pos_ >= data_.size()) return -1;
    const size_t m = std::min(len, data_.size() - pos_);
    memcpy(b, data_.data() + pos_, m);
    pos_ += m;
    return (long)m;
}
Bytes ByteArrayInputStream::readNBytes(size_t n) {
    const size_t m = std::min(n, data_.size() - pos_);
    Bytes out(data_.begin() + (long)pos_, data_.begin() + (long)(pos_ + m));
    pos_ += m;
    return out;
}
Bytes ByteArrayInputStream::readAllBytes() { return readNBytes(data_.size() - pos_); }

IvSupplier secureRandomIvSupplier() {
    return [](uint8_t iv[IV_SIZE]) {
        FILE* f = fopen("/dev/urandom", "rb");
        if (!f || fread(iv, 1, IV_SIZE, f) != IV_SIZE) { if (f) fclose(f); throw std::runtime_error("cannot read /dev/urandom"); }
        fclose(f);
    };
}

// =====================================================================================================
// upload side
// =====================================================================================================
BaseTransformChunkEnumeration::BaseTransformChunkEnumeration(std::shared_ptr<InputStream> in, int originalChunkSize) : in_(std::move(in)) {
    if (!in_) throw std::invalid_argument("inputStream cannot be null");
    if (originalChunkSize < 0) throw std::invalid_argument("originalChunkSize must be non-negative, " + std::to_string(originalChunkSize) + " given");
    originalChunkSize_ = originalChunkSize;
}
void BaseTransformChunkEnumeration::fillChunkIfNeeded() {                            // BaseTransformChunkEnumeration.java:79-93
    if (chunk_) return;
    chunk_ = originalChunkSize_ != 0 ? in_->readNBytes((size_t)originalChunkSize_) : in_->readAllBytes();
}
bool BaseTransformChunkEnumeration::hasMoreElements() { fillChunkIfNeeded(); return !chunk_->empty(); }
Bytes BaseTransformChunkEnumeration::nextElement() {
    fillChunkIfNeeded();
    if (chunk_->empty()) throw std::out_of_range("NoSuchElementException");
    Bytes r = std::move(*chunk_);
    chunk_.reset();
    return r;
}

GpuTransformChunkEnumeration::GpuTransformChunkEnumeration(std::shared_ptr<Backend> be, std::shared_ptr<TransformChunkEnumeration> inner, bool compress,
                                                           std::optional<DataKeyAndAAD> enc, IvSupplier iv, int batchChunks, bool withCrc, uint32_t profile, bool readAhead)
    : be_(std::move(be)), inner_(std::move(inner)), compress_(compress), enc_(std::move(enc)), iv_(std::move(iv)), batch_(batchChunks), withCrc_(withCrc),
      profile_(profile), readAhead_(readAhead) {
    if (!inner_) throw std::invalid_argument("inner cannot be null");
    if (batch_ < 1) throw std::invalid_argument("batchChunks must be positive");
    if (enc_ && enc_->dataKey.size() != 32) throw std::invalid_argument("AES-256 data key must be 32 bytes");
    // CompressionChunkEnumeration.java:39-42 (null) then EncryptionChunkEnumeration.java:41-47 (inner + ivSize + getOutputSize)
    const std::optional<int> innerSize = inner_->transformedChunkSize();
    if (compress_ || !innerSize) transformedChunkSize_ = std::nullopt;
    else transformedChunkSize_ = enc_ ? *innerSize + IV_SIZE + GCM_TAG_BYTES : *innerSize;
}
GpuTransformChunkEnumeration::~GpuTransformChunkEnumeration() {
    if (ahead_.valid()) { try { ahead_.get(); } catch (...) {} }       // the helper uses this object: it is gone before the members are
}

GpuTransformChunkEnumeration::Batch GpuTransformChunkEnumeration::transformNextBatch() {
    Batch out;
    std::vector<Bytes> in;
    while ((int)in.size() < batch_ && inner_->hasMoreElements()) in.push_back(inner_->nextElement());
    if (in.empty()) return out;
    const uint32_t flags = (compress_ ? TSX_COMPRESS : 0u) | (enc_ ? TSX_ENCRYPT : 0u) | (withCrc_ ? TSX_CRC : 0u);
    if ((flags & (TSX_COMPRESS | TSX_ENCRYPT)) == 0 && !withCrc_) { out.chunks = std::move(in); return out; }     // pure base: nothing to do
    std::vector<tsx_chunk_desc> d(in.size());
    size_t so = 0, dofs = 0;
    for (size_t i = 0; i < in.size(); i++) {
        memset(&d[i], 0, sizeof d[i]);
        d[i].src_off = so; d[i].src_len = (uint32_t)in[i].size(); so += align16(in[i].size()) + 16;
        const size_t cap = be_->transformedBound(in[i].size(), flags);
        d[i].dst_off = dofs; d[i].dst_cap = (uint32_t)cap; dofs += align16(cap) + 16;
        if (enc_) iv_(d[i].iv);
    }
    Bytes src(so + 16), dst(dofs + 16);
    for (size_t i = 0; i < in.size(); i++) memcpy(src.data() + d[i].src_off, in[i].data(), in[i].size());
    const tsx_batch_params p = makeParams(flags, enc_ ? &enc_->dataKey : nullptr, enc_ ? &enc_->aad : nullptr, profile_);
    be_->transformBatch(p, d, src.data(), src.size(), dst.data(), dst.size());
    for (size_t i = 0; i < in.size(); i++) {
        if (d[i].status != TSX_OK) throw std::runtime_error(be_->strerror(d[i].status));        // the reference wraps crypto failures in RuntimeException
        if (withCrc_) out.crcs.push_back(d[i].crc32c);
        out.chunks.emplace_back(dst.begin() + (long)d[i].dst_off, dst.begin() + (long)(d[i].dst_off + d[i].dst_len));
    }
    return out;
}

void GpuTransformChunkEnumeration::fillBatchIfNeeded() {
    if (next_ < ready_.size() || exhausted_) return;
    ready_.clear(); next_ = 0;
    // the batch the helper has been working on (its failure surfaces here, where its first chunk is asked for), else a fresh one
    Batch b = ahead_.valid() ? ahead_.get() : transformNextBatch();
    if (b.chunks.empty()) { exhausted_ = true; return; }
    ready_ = std::move(b.chunks);
    crcs_.insert(crcs_.end(), b.crcs.begin(), b.crcs.end());
    if (readAhead_) ahead_ = std::async(std::launch::async, [this] { return transformNextBatch(); });
}
GpuTransformChunkEnumeration::PackedBatch GpuTransformChunkEnumeration::transformNextBatchPacked() {
    PackedBatch out;
    std::vector<Bytes> in;
    while ((int)in.size() < batch_ && inner_->hasMoreElements()) in.push_back(inner_->nextElement());
    if (in.empty()) return out;
    const uint32_t flags = (compress_ ? TSX_COMPRESS : 0u) | (enc_ ? TSX_ENCRYPT : 0u) | (withCrc_ ? TSX_CRC : 0u);
    if ((flags & (TSX_COMPRESS | TSX_ENCRYPT)) == 0) {                   // pure base: the chunks are the object
        for (const Bytes& c : in) {
            out.object.insert(out.object.end(), c.begin(), c.end());
            out.sizes.push_back((int)c.size());
            if (withCrc_) out.crcs.push_back(be_->crc32c(c.data(), c.size()));
        }
        return out;
    }
    std::vector<tsx_chunk_desc> d(in.size());
    size_t so = 0, bound = 0;
    for (size_t i = 0; i < in.size(); i++) {
        memset(&d[i], 0, sizeof d[i]);
        d[i].src_off = so; d[i].src_len = (uint32_t)in[i].size(); so += align16(in[i].size()) + 16;
        bound += be_->transformedBound(in[i].size(), flags);
        if (enc_) iv_(d[i].iv);
    }
    Bytes src(so + 16);
    for (size_t i = 0; i < in.size(); i++) if (!in[i].empty()) memcpy(src.data() + d[i].src_off, in[i].data(), in[i].size());
    const tsx_batch_params p = makeParams(flags, enc_ ? &enc_->dataKey : nullptr, enc_ ? &enc_->aad : nullptr, profile_);
    out.object.resize(bound);                                          // room for the worst case, trimmed to what was produced
    be_->transformBatchPacked(p, d, src.data(), src.size(), out.object.data(), bound);
    size_t total = 0;
    for (size_t i = 0; i < in.size(); i++) {
        if (d[i].status != TSX_OK) throw std::runtime_error(be_->strerror(d[i].status));
        if (withCrc_) out.crcs.push_back(d[i].crc32c);
        out.sizes.push_back((int)d[i].dst_len); total += d[i].dst_len;
    }
    out.object.resize(total);
    return out;
}
size_t GpuTransformChunkEnumeration::appendNextBatchPacked(Bytes& object, std::vector<int>& sizes) {
    if (next_ < ready_.size() || ahead_.valid()) throw std::logic_error("appendNextBatchPacked after nextElement inside a batch");
    PackedBatch b = transformNextBatchPacked();
    if (b.sizes.empty()) return 0;
    if (object.empty()) object = std::move(b.object); else object.insert(object.end(), b.object.begin(), b.object.end());
    sizes.insert(sizes.end(), b.sizes.begin(), b.sizes.end());
    crcs_.insert(crcs_.end(), b.crcs.begin(), b.crcs.end());
    size_t total = 0; for (int v : b.sizes) total += (size_t)v;
    return total;
}
bool GpuTransformChunkEnumeration::hasMoreElements() { fillBatchIfNeeded(); return next_ < ready_.size(); }
Bytes GpuTransformChunkEnumeration::nextElement() {
    fillBatchIfNeeded();
    if (next_ >= ready_.size()) throw std::out_of_range("NoSuchElementException");
    return std::move(ready_[next_++]);
}

TransformFinisher::TransformFinisher(std::shared_ptr<TransformChunkEnumeration> inner, int originalFileSize, bool chunkingEnabled)
    : inner_(std::move(inner)), originalFileSize_(originalFileSize) {
    if (!inner_) throw std::invalid_argument("inner cannot be null");
    if (originalFileSize < 0) throw std::invalid_argument("originalFileSize must be non-negative, " + std::to_string(originalFileSize) + " given");
    const int originalChunkSize = chunkingEnabled ? inner_->originalChunkSize() : originalFileSize;      // TransformFinisher.java:68
    const std::optional<int> t = inner_->transformedChunkSize();                                          // :75-93
    if (!t) builder_.reset(new VariableSizeChunkIndexBuilder(originalChunkSize, originalFileSize));
    else builder_.reset(new FixedSizeChunkIndexBuilder(originalChunkSize, originalFileSize, *t));
}
bool TransformFinisher::isBaseTransform() const { return dynamic_cast<BaseTransformChunkEnumeration*>(inner_.get()) != nullptr; }
Bytes TransformFinisher::nextElement() {                                              // TransformFinisher.java:101-110
    Bytes chunk = inner_->nextElement();
    if (hasMoreElements()) builder_->addChunk((int)chunk.size());
    else chunkIndex_ = builder_->finish((int)chunk.size());
    return chunk;
}
std::shared_ptr<ChunkIndex> TransformFinisher::chunkIndex() {                          // :112-132
    if (!chunkIndex_) {
        if (!isBaseTransform()) throw std::logic_error("Chunk index was not built, was finisher used?");
        const int chunkSize = *inner_->transformedChunkSize();
        int size = originalFileSize_;
        while (size > chunkSize) { builder_->addChunk(chunkSize); size -= chunkSize; }
        chunkIndex_ = builder_->finish(size);
    }
    return chunkIndex_;
}
Bytes TransformFinisher::toBytesPacked() {
    auto* gpu = dynamic_cast<GpuTransformChunkEnumeration*>(inner_.get());
    if (!gpu) return toBytes();
    Bytes out;
    std::vector<int> sizes;
    while (gpu->appendNextBatchPacked(out, sizes)) {}
    // the index sees the same sizes in the same order as through nextElement() (TransformFinisher.java:101-110)
    if (sizes.empty()) return out;                                     // nothing passed through: like toBytes(), no index is built
    for (size_t i = 0; i + 1 < sizes.size(); i++) builder_->addChunk(sizes[i]);
    chunkIndex_ = builder_->finish(sizes.back());
    return out;
}
Bytes TransformFinisher::toBytes() {
    Bytes out;
    while (hasMoreElements()) { const Bytes c = nextElement(); out.insert(out.end(), c.begin(), c.end()); }
    return out;
}

// ---- GpuTransformFinisher (Java: java/io/aiven/kafka/tieredstorage/gpu/GpuTransformFinisher.java) ------------------------
GpuTransformFinisher::GpuTransformFinisher(std::shared_ptr<GpuTransformChunkEnumeration> inner, int originalFileSize, bool chunkingEnabled,
                                           std::shared_ptr<TokenBucket> rateLimitingBucket, bool readAhead)
    : inner_(std::move(inner)), bucket_(std::move(rateLimitingBucket)), readAhead_(readAhead) {
    if (!inner_) throw std::invalid_argument("inner cannot be null");
    if (originalFileSize < 0) throw std::invalid_argument("originalFileSize must be non-negative, " + std::to_string(originalFileSize) + " given");
    const int originalChunkSize = chunkingEnabled ? inner_->originalChunkSize() : originalFileSize;      // TransformFinisher.java:68
    const std::optional<int> t = inner_->transformedChunkSize();                                          // :75-93
    if (!t) builder_.reset(new VariableSizeChunkIndexBuilder(originalChunkSize, originalFileSize));
    else builder_.reset(new FixedSizeChunkIndexBuilder(originalChunkSize, originalFileSize, *t));
}
GpuTransformFinisher::~GpuTransformFinisher() {
    if (ahead_.valid()) { try { ahead_.get(); } catch (...) {} }       // the helper uses inner_: it is gone before the members are
}
// The next packed batch becomes the one being read.  Sizes go to the index builder in order, the newest one held back: the reference
// calls addChunk for every chunk but the object's last and finish for that one (TransformFinisher.java:101-110), and which chunk is
// the last is only known when the batch behind it comes back empty.
bool GpuTransformFinisher::nextBatch() {
    if (exhausted_) return false;
    GpuTransformChunkEnumeration::PackedBatch b = ahead_.valid() ? ahead_.get() : inner_->transformNextBatchPacked();
    if (b.sizes.empty()) {
        exhausted_ = true;
        if (pending_) { chunkIndex_ = builder_->finish(*pending_); pending_.reset(); }
        cur_.clear(); pos_ = 0;
        return false;
    }
    for (int v : b.sizes) {
        if (pending_) builder_->addChunk(*pending_);
        pending_ = v;
    }
    crcs_.insert(crcs_.end(), b.crcs.begin(), b.crcs.end());
    cur_ = std::move(b.object); pos_ = 0;
    if (readAhead_) ahead_ = std::async(std::launch::async, [this] { return inner_->transformNextBatchPacked(); });
    return true;
}
size_t GpuTransformFinisher::fillPart(uint8_t* part, size_t capacity) {
    size_t at = 0;
    while (at < capacity) {
        if (pos_ >= cur_.size() && !nextBatch()) break;
        const size_t m = std::min(capacity - at, cur_.size() - pos_);
        memcpy(part + at, cur_.data() + pos_, m);
        pos_ += m; at += m;
    }
    return at;
}
namespace {
class PackedObjectStream : public InputStream {
public:
    explicit PackedObjectStream(GpuTransformFinisher* f) : f_(f) {}
    long read(uint8_t* b, size_t len) override {                       // like SequenceInputStream: at most what the current batch still holds
        if (len == 0) return 0;
        const size_t m = f_->readSome(b, len);
        return m ? (long)m : -1;
    }

private:
    GpuTransformFinisher* f_;
};
}  // namespace
size_t GpuTransformFinisher::readSome(uint8_t* b, size_t len) {
    while (pos_ >= cur_.size()) if (!nextBatch()) return 0;
    const size_t m = std::min(len, cur_.size() - pos_);
    memcpy(b, cur_.data() + pos_, m);
    pos_ += m;
    return m;
}
std::shared_ptr<InputStream> GpuTransformFinisher::toInputStream() {
    std::shared_ptr<InputStream> s = std::make_shared<PackedObjectStream>(this);
    if (bucket_) s = std::make_shared<RateLimitedInputStream>(s, bucket_);     // TransformFinisher.java:146-151
    return s;
}
std::shared_ptr<ChunkIndex> GpuTransformFinisher::chunkIndex() {
    if (!chunkIndex_) throw std::logic_error("Chunk index was not built, was finisher used?");      // TransformFinisher.java:112-122
    return chunkIndex_;
}

// =====================================================================================================
// fetch side
// =====================================================================================================
BaseDetransformChunkEnumeration::BaseDetransformChunkEnumeration(std::shared_ptr<InputStream> in) : in_(std::move(in)), isEmpty_(true) {
    if (!in_) throw std::invalid_argument("inputStream cannot be null");
}
BaseDetransformChunkEnumeration::BaseDetransformChunkEnumeration(std::shared_ptr<InputStream> in, std::vector<Chunk> chunks)
    : in_(std::move(in)), chunks_(std::move(chunks)), isEmpty_(chunks_.empty()) {
    if (!in_) throw std::invalid_argument("inputStream cannot be null");
}
void BaseDetransformChunkEnumeration::fillChunkIfNeeded() {                            // BaseDetransformChunkEnumeration.java:78-115
    if (chunk_) return;
    if (iter_ >= chunks_.size() && !isEmpty_) {
        chunk_ = Bytes{};
        if (!inputStreamClosed_) { in_->close(); inputStreamClosed_ = true; }
        return;
    }
    if (inputStreamClosed_) throw std::runtime_error("Input stream already closed");
    if (!isEmpty_) {
        const int expected = chunks_[iter_++].transformedSize;
        chunk_ = in_->readNBytes((size_t)expected);
        if ((int)chunk_->size() < expected) throw std::runtime_error("Stream has fewer bytes than expected");
    } else {
        chunk_ = in_->readAllBytes();
    }
}
bool BaseDetransformChunkEnumeration::hasMoreElements() { fillChunkIfNeeded(); return !chunk_->empty(); }
Bytes BaseDetransformChunkEnumeration::nextElement() {
    fillChunkIfNeeded();
    if (chunk_->empty()) throw std::out_of_range("NoSuchElementException");
    Bytes r = std::move(*chunk_);
    chunk_.reset();
    return r;
}

GpuDetransformChunkEnumeration::GpuDetransformChunkEnumeration(std::shared_ptr<Backend> be, std::shared_ptr<DetransformChunkEnumeration> inner, bool compressed,
                                                               std::optional<SegmentEncryptionMetadata> enc, int maxOriginalChunkSize, int batchChunks)
    : be_(std::move(be)), inner_(std::move(inner)), compressed_(compressed), enc_(std::move(enc)), maxOriginal_(maxOriginalChunkSize), batch_(batchChunks) {
    if (!inner_) throw std::invalid_argument("inner cannot be null");
    if (batch_ < 1) throw std::invalid_argument("batchChunks must be positive");
}

void GpuDetransformChunkEnumeration::fillBatchIfNeeded() {
    if (next_ < ready_.size()) return;
    ready_.clear(); next_ = 0;
    std::vector<Bytes> in;
    while ((int)in.size() < batch_ && inner_->hasMoreElements()) in.push_back(inner_->nextElement());
    if (in.empty()) return;
    const uint32_t flags = (compressed_ ? TSX_COMPRESS : 0u) | (enc_ ? TSX_ENCRYPT : 0u);
    if (flags == 0) { ready_ = std::move(in); return; }
    std::vector<tsx_chunk_desc> d(in.size());
    size_t so = 0, dofs = 0;
    for (size_t i = 0; i < in.size(); i++) {
        memset(&d[i], 0, sizeof d[i]);
        d[i].src_off = so; d[i].src_len = (uint32_t)in[i].size(); so += align16(in[i].size()) + 16;
        size_t cap;
        if (compressed_) {
            // the frame states its content size (the reference asks Zstd.decompressedSize first); an encrypted frame is not
            // readable on the host, so the slot is the original chunk size, which bounds every chunk of the segment
            cap = (size_t)maxOriginal_;
            if (!enc_) { const long long n = zstdFrameContentSize(in[i]); if (n >= 0) cap = (size_t)n; }
        } else {
            cap = in[i].size() >= (size_t)(IV_SIZE + GCM_TAG_BYTES) ? in[i].size() - IV_SIZE - GCM_TAG_BYTES : 0;
        }
        d[i].dst_off = dofs; d[i].dst_cap = (uint32_t)cap; dofs += align16(cap) + 16;
    }
    Bytes src(so + 16), dst(dofs + 16);
    for (size_t i = 0; i < in.size(); i++) memcpy(src.data() + d[i].src_off, in[i].data(), in[i].size());
    const tsx_batch_params p = makeParams(flags, enc_ ? &enc_->dataKey : nullptr, enc_ ? &enc_->aad : nullptr, TSX_ZSTD_PROFILE_1_5_7);
    be_->detransformBatch(p, d, src.data(), src.size(), dst.data(), dst.size());
    for (size_t i = 0; i < in.size(); i++) {
        if (d[i].status != TSX_OK) {
            // the failure belongs to chunk i: everything before it is still handed out, the exception surfaces when i is reached
            std::string msg = be_->strerror(d[i].status);
            if (d[i].status == TSX_E_BAD_SIZE) msg = "Invalid decompressed size: " + std::to_string(zstdFrameContentSize(in[i]));
            failMsg_ = msg;
            return;
        }
        ready_.emplace_back(dst.begin() + (long)d[i].dst_off, dst.begin() + (long)(d[i].dst_off + d[i].dst_len));
    }
}
bool GpuDetransformChunkEnumeration::hasMoreElements() {
    if (next_ >= ready_.size() && !failMsg_) fillBatchIfNeeded();
    return next_ < ready_.size() || failMsg_.has_value();
}
Bytes GpuDetransformChunkEnumeration::nextElement() {
    if (next_ >= ready_.size() && !failMsg_) fillBatchIfNeeded();
    if (next_ < ready_.size()) return std::move(ready_[next_++]);
    if (failMsg_) throw std::runtime_error(*failMsg_);
    throw std::out_of_range("NoSuchElementException");
}

Bytes DetransformFinisher::toBytes() {                                                // DetransformFinisher.java:48-54
    if (auto* base = dynamic_cast<BaseDetransformChunkEnumeration*>(inner_.get())) return base->inputStream()->readAllBytes();
    Bytes out;
    while (inner_->hasMoreElements()) { const Bytes c = inner_->nextElement(); out.insert(out.end(), c.begin(), c.end()); }
    return out;
}

Bytes GpuChunkManager::getChunk(const std::string& objectKey, const SegmentManifest& manifest, int chunkId) {      // DefaultChunkManager.java:50-70
    return getChunks(objectKey, manifest, chunkId, 1)[0];
}

std::vector<Bytes> GpuChunkManager::getChunks(const std::string& objectKey, const SegmentManifest& manifest, int firstChunkId, int count) {
    const std::vector<Chunk>& all = manifest.chunkIndex->chunks();
    if (firstChunkId < 0 || count < 1 || (size_t)(firstChunkId + count) > all.size()) throw std::out_of_range("chunk id out of range");
    std::vector<Chunk> wanted(all.begin() + firstChunkId, all.begin() + firstChunkId + count);
    const BytesRange range{wanted.front().transformedPosition, wanted.back().transformedPosition + wanted.back().transformedSize - 1};
    std::shared_ptr<InputStream> content = fetcher_->fetch(objectKey, range);
    std::shared_ptr<DetransformChunkEnumeration> e = std::make_shared<BaseDetransformChunkEnumeration>(content, wanted);
    if (manifest.encryption || manifest.compression)
        e = std::make_shared<GpuDetransformChunkEnumeration>(be_, e, manifest.compression, manifest.encryption, manifest.chunkIndex->originalChunkSize(), count);
    std::vector<Bytes> out;
    while (e->hasMoreElements()) out.push_back(e->nextElement());
    return out;
}


// =====================================================================================================
// SegmentManifestV1 JSON (manifest/SegmentManifestV1.java:30-132 through Jackson; goldens SegmentManifestV1SerdeTest.java:82-133)
// =====================================================================================================
namespace {
// One level of a JSON object: (key, raw text of the value) in document order.  Strings may contain escapes; nesting is skipped by
// depth counting outside strings.
std::vector<std::pair<std::string, std::string>> jsonFields(const std::string& j) {
    std::vector<std::pair<std::string, std::string>> out;
    size_t i = 0;
    auto ws = [&] { while (i < j.size() && (j[i] == ' ' || j[i] == '\n' || j[i] == '\t' || j[i] == '\r')) i++; };
    auto str = [&]() -> std::string {
        if (j[i] != '"') throw std::invalid_argument("malformed JSON: expected a string");
        size_t b = ++i;
        while (i < j.size() && j[i] != '"') i += (j[i] == '\\') ? 2 : 1;
        if (i >= j.size()) throw std::invalid_argument("malformed JSON: unterminated string");
        return j.substr(b, i++ - b);
    };
    ws();
    if (i >= j.size() || j[i] != '{') throw std::invalid_argument("malformed JSON: expected an object");
    i++; ws();
    if (i < j.size() && j[i] == '}') return out;
    for (;;) {
        ws();
        std::string key = str();
        ws();
        if (i >= j.size() || j[i] != ':') throw std::invalid_argument("malformed JSON: expected ':'");
        i++; ws();
        const size_t b = i;
        int depth = 0;
        for (; i < j.size(); i++) {
            const char c = j[i];
            if (c == '"') { i++; while (i < j.size() && j[i] != '"') i += (j[i] == '\\') ? 2 : 1; continue; }
            if (c == '{' || c == '[') depth++;
            else if (c == '}' || c == ']') { if (depth == 0) break; depth--; }
            else if (c == ',' && depth == 0) break;
        }
        if (i >= j.size()) throw std::invalid_argument("malformed JSON: unterminated object");
        size_t e = i;
        while (e > b && (j[e - 1] == ' ' || j[e - 1] == '\n' || j[e - 1] == '\t' || j[e - 1] == '\r')) e--;
        out.emplace_back(std::move(key), j.substr(b, e - b));
        if (j[i] == '}') break;
        i++;
    }
    return out;
}
const std::string* findField(const std::vector<std::pair<std::string, std::string>>& f, const char* name) {
    for (const auto& kv : f) if (kv.first == name) return &kv.second;
    return nullptr;
}
const std::string& needField(const std::vector<std::pair<std::string, std::string>>& f, const char* name) {
    const std::string* v = findField(f, name);
    if (!v) throw std::invalid_argument(std::string("Missing required creator property '") + name + "'");
    return *v;
}
std::string unquote(const std::string& raw) {
    if (raw.size() < 2 || raw.front() != '"' || raw.back() != '"') throw std::invalid_argument("malformed JSON: expected a string value");
    return raw.substr(1, raw.size() - 2);
}
std::string indexJson(const SegmentIndexV1& x) { return "{\"position\":" + std::to_string(x.position) + ",\"size\":" + std::to_string(x.size) + "}"; }
SegmentIndexV1 indexFromJson(const std::string& raw) {
    const auto f = jsonFields(raw);
    return SegmentIndexV1{std::stoi(needField(f, "position")), std::stoi(needField(f, "size"))};
}
}  // namespace

namespace {
const char* indexTypeName(IndexType t) {
    switch (t) {
        case IndexType::OFFSET: return "OFFSET"; case IndexType::TIMESTAMP: return "TIMESTAMP"; case IndexType::PRODUCER_SNAPSHOT: return "PRODUCER_SNAPSHOT";
        case IndexType::TRANSACTION: return "TRANSACTION"; default: return "LEADER_EPOCH";
    }
}
}  // namespace

SegmentIndexesV1Builder& SegmentIndexesV1Builder::add(IndexType type, int size) {
    for (const auto& a : added_) if (a.first == type) throw std::logic_error(std::string("Index ") + indexTypeName(type) + " is already added");
    added_.emplace_back(type, SegmentIndexV1{currentPosition_, size});
    currentPosition_ += size;
    return *this;
}
std::vector<IndexType> SegmentIndexesV1Builder::indexes() const {
    std::vector<IndexType> v;
    for (const auto& a : added_) v.push_back(a.first);
    std::sort(v.begin(), v.end());
    return v;
}
SegmentIndexesV1 SegmentIndexesV1Builder::build() const {
    auto has = [&](IndexType t) { for (const auto& a : added_) if (a.first == t) return true; return false; };
    auto get = [&](IndexType t) { for (const auto& a : added_) if (a.first == t) return a.second; return SegmentIndexV1{}; };
    if (added_.size() < 4) {
        std::string list;
        for (IndexType t : indexes()) { if (!list.empty()) list += ", "; list += indexTypeName(t); }
        throw std::logic_error("Not enough indexes have been added; at least 4 required. Indexes included: [" + list + "]");
    }
    if (added_.size() == 4 && has(IndexType::TRANSACTION)) throw std::logic_error("OFFSET, TIMESTAMP, PRODUCER_SNAPSHOT, and LEADER_EPOCH indexes are required");
    SegmentIndexesV1 out;
    out.offset = get(IndexType::OFFSET); out.timestamp = get(IndexType::TIMESTAMP); out.producerSnapshot = get(IndexType::PRODUCER_SNAPSHOT);
    out.leaderEpoch = get(IndexType::LEADER_EPOCH);
    if (has(IndexType::TRANSACTION)) out.transaction = get(IndexType::TRANSACTION);
    return out;
}

std::string segmentManifestToJson(Backend& be, const SegmentManifestV1& m, const DataKeyEncryptor& wrapKey) {
    if (!m.chunkIndex) throw std::invalid_argument("chunkIndex cannot be null");
    std::string j = "{\"version\":\"1\",\"chunkIndex\":" + chunkIndexToJson(be, *m.chunkIndex) + ",\"segmentIndexes\":{";
    const SegmentIndexesV1& si = m.segmentIndexes;
    j += "\"offset\":" + indexJson(si.offset) + ",\"timestamp\":" + indexJson(si.timestamp) + ",\"producerSnapshot\":" + indexJson(si.producerSnapshot) +
         ",\"leaderEpoch\":" + indexJson(si.leaderEpoch) + ",\"transaction\":" + (si.transaction ? indexJson(*si.transaction) : std::string("null")) + "}";
    j += std::string(",\"compression\":") + (m.compression ? "true" : "false");
    if (m.encryption) {                                                // @JsonInclude(NON_ABSENT): no property at all without encryption
        if (!wrapKey) throw std::invalid_argument("a data-key encryptor is required to serialise an encrypted segment's manifest");
        j += ",\"encryption\":{\"dataKey\":\"" + wrapKey(m.encryption->dataKey) + "\",\"aad\":\"" + base64Encode(m.encryption->aad) + "\"}";
    }
    if (!m.remoteLogSegmentMetadataJson.empty()) j += ",\"remoteLogSegmentMetadata\":" + m.remoteLogSegmentMetadataJson;
    return j + "}";
}

SegmentManifestV1 segmentManifestFromJson(Backend& be, const std::string& json, const DataKeyDecryptor& unwrapKey) {
    const auto f = jsonFields(json);
    const std::string version = unquote(needField(f, "version"));
    if (version != "1") throw std::invalid_argument("Could not resolve type id '" + version + "' as a subtype of SegmentManifest");
    SegmentManifestV1 m;
    m.chunkIndex = chunkIndexFromJson(be, needField(f, "chunkIndex"));
    const auto si = jsonFields(needField(f, "segmentIndexes"));
    m.segmentIndexes.offset = indexFromJson(needField(si, "offset"));
    m.segmentIndexes.timestamp = indexFromJson(needField(si, "timestamp"));
    m.segmentIndexes.producerSnapshot = indexFromJson(needField(si, "producerSnapshot"));
    m.segmentIndexes.leaderEpoch = indexFromJson(needField(si, "leaderEpoch"));
    const std::string& txn = needField(si, "transaction");            // required = true, value may be null
    if (txn != "null") m.segmentIndexes.transaction = indexFromJson(txn);
    const std::string& comp = needField(f, "compression");
    if (comp != "true" && comp != "false") throw std::invalid_argument("compression must be a boolean");
    m.compression = comp == "true";
    if (const std::string* enc = findField(f, "encryption")) {
        if (*enc != "null") {
            const auto ef = jsonFields(*enc);
            if (!unwrapKey) throw std::invalid_argument("a data-key decryptor is required to read an encrypted segment's manifest");
            SegmentEncryptionMetadata e;
            e.dataKey = unwrapKey(unquote(needField(ef, "dataKey")));
            e.aad = base64Decode(unquote(needField(ef, "aad")));
            m.encryption = e;
        }
    }
    // remoteLogSegmentMetadata: written for humans and tooling, never read back (JsonProperty.Access.READ_ONLY); kept verbatim
    if (const std::string* r = findField(f, "remoteLogSegmentMetadata")) m.remoteLogSegmentMetadataJson = *r;
    return m;
}

// =====================================================================================================
// GpuChunkCache
// =====================================================================================================
struct GpuChunkCache::Impl {
    using Key = std::pair<std::string, int>;
    struct Batch {                                                     // one getChunks call being put together / in flight
        std::string object; int first = 0, count = 0;
        bool open = true;                                              // still accepts adjacent chunks
        std::vector<std::shared_ptr<std::promise<Bytes>>> slots;
    };
    std::shared_ptr<GpuChunkManager> mgr;
    int prefetchingSize; size_t maxBytes; int getTimeoutMs, coalesceWaitMicros;
    mutable std::mutex mu;
    std::list<std::pair<Key, Bytes>> lru;                              // front = most recently used
    std::map<Key, std::list<std::pair<Key, Bytes>>::iterator> cached;
    std::map<Key, std::shared_future<Bytes>> pending;
    std::map<std::string, std::shared_ptr<Batch>> openBatch;           // per object: the batch that can still grow at its end
    std::vector<std::future<void>> helpers;                           // prefetch batches nobody is waiting for yet
    size_t bytes = 0;
    ChunkCacheStats st;

    void insert(const Key& k, const Bytes& v) {                        // under mu
        auto it = cached.find(k);
        if (it != cached.end()) { bytes -= it->second->second.size(); lru.erase(it->second); cached.erase(it); }
        lru.emplace_front(k, v); cached[k] = lru.begin(); bytes += v.size();
        while (bytes > maxBytes && lru.size() > 1) { bytes -= lru.back().second.size(); cached.erase(lru.back().first); lru.pop_back(); st.evictions++; }
    }
    // Runs one batch: one ranged fetch + one device batch; on failure every chunk on its own, so that only the bad one fails.
    // Whatever happens in here - a manager that throws, returns too few chunks, an allocation that fails while the chunk is cached - no
    // slot of the batch stays open and no key stays in `pending` (GpuChunkCache.java runBatch / abandon: later getChunk calls for these
    // ids must start a new load, not wait get.timeout.ms for a task that died).
    void runBatch(const std::shared_ptr<Batch>& b, const SegmentManifest& manifest) {
        std::vector<char> settled((size_t)b->count, 0);
        std::exception_ptr fatal;
        try {
            std::vector<Bytes> got;
            std::exception_ptr batchErr;
            try {
                got = mgr->getChunks(b->object, manifest, b->first, b->count);
                if ((int)got.size() != b->count) throw std::runtime_error("getChunks returned " + std::to_string(got.size()) + " chunks for " + std::to_string(b->count));
            } catch (...) { batchErr = std::current_exception(); }
            { std::lock_guard<std::mutex> lk(mu); st.fetchCalls++; st.chunksFetched += b->count; }
            for (int i = 0; i < b->count; i++) {
                const Key k{b->object, b->first + i};
                Bytes v;
                std::exception_ptr err;
                if (!batchErr) v = std::move(got[(size_t)i]);
                else if (b->count == 1) err = batchErr;
                else { try { v = mgr->getChunk(b->object, manifest, b->first + i); } catch (...) { err = std::current_exception(); } }
                { std::lock_guard<std::mutex> lk(mu); if (!err) insert(k, v); pending.erase(k); }
                settled[(size_t)i] = 1;
                if (err) b->slots[(size_t)i]->set_exception(err); else b->slots[(size_t)i]->set_value(std::move(v));
            }
        } catch (...) { fatal = std::current_exception(); }
        for (int i = 0; i < b->count; i++) {
            if (settled[(size_t)i]) continue;
            { std::lock_guard<std::mutex> lk(mu); pending.erase(Key{b->object, b->first + i}); }
            try { b->slots[(size_t)i]->set_exception(fatal ? fatal : std::make_exception_ptr(std::runtime_error("chunk load ended without a result"))); }
            catch (const std::future_error&) {}                          // (the slot was settled by the statement that threw)
        }
    }
};

GpuChunkCache::GpuChunkCache(std::shared_ptr<GpuChunkManager> manager, int prefetchingSize, size_t maxBytes, int getTimeoutMs, int coalesceWaitMicros)
    : impl_(new Impl) {
    impl_->mgr = std::move(manager); impl_->prefetchingSize = prefetchingSize; impl_->maxBytes = maxBytes;
    impl_->getTimeoutMs = getTimeoutMs; impl_->coalesceWaitMicros = coalesceWaitMicros;
}
GpuChunkCache::~GpuChunkCache() { quiesce(); }
void GpuChunkCache::quiesce() {
    std::vector<std::future<void>> hs;
    { std::lock_guard<std::mutex> lk(impl_->mu); hs.swap(impl_->helpers); }
    for (auto& h : hs) if (h.valid()) h.wait();
}
ChunkCacheStats GpuChunkCache::stats() const { std::lock_guard<std::mutex> lk(impl_->mu); return impl_->st; }

Bytes GpuChunkCache::getChunk(const std::string& objectKey, const SegmentManifest& manifest, int chunkId) {
    Impl& I = *impl_;
    const std::vector<Chunk>& all = manifest.chunkIndex->chunks();
    if (chunkId < 0 || (size_t)chunkId >= all.size()) throw std::out_of_range("chunk id out of range");
    // the window this call may touch: the chunk itself + ChunkCache.startPrefetching's range behind it (never more)
    int last = chunkId;
    if (I.prefetchingSize > 0) {
        const Chunk& cur = all[(size_t)chunkId];
        const long long start = (long long)cur.originalPosition + cur.originalSize;
        const long long end = std::min<long long>(start + I.prefetchingSize - 1, 2147483647LL);
        if (start <= end) for (const Chunk& c : manifest.chunkIndex->chunksForRange(BytesRange{(int)start, (int)end})) last = std::max(last, c.id);
    }
    std::shared_future<Bytes> mine;
    std::shared_ptr<Impl::Batch> lead;                                 // a batch this caller has to send off
    {
        std::unique_lock<std::mutex> lk(I.mu);
        const Impl::Key k{objectKey, chunkId};
        auto hit = I.cached.find(k);
        bool have = false;
        Bytes cachedCopy;
        if (hit != I.cached.end()) { I.st.hits++; I.lru.splice(I.lru.begin(), I.lru, hit->second); cachedCopy = hit->second->second; have = true; }
        else if (I.pending.count(k)) { I.st.hits++; mine = I.pending[k]; }
        else I.st.misses++;
        // what is neither cached nor on its way, as runs of consecutive ids; the run containing chunkId is waited for, the rest is prefetch
        int i = chunkId;
        while (i <= last) {
            while (i <= last && (I.cached.count({objectKey, i}) || I.pending.count({objectKey, i}))) i++;
            if (i > last) break;
            int j = i;
            while (j <= last && !I.cached.count({objectKey, j}) && !I.pending.count({objectKey, j})) j++;
            // join the batch that is about to leave for this object if the run continues it, else open a new one
            std::shared_ptr<Impl::Batch> b;
            auto ob = I.openBatch.find(objectKey);
            const bool joins = ob != I.openBatch.end() && ob->second->open && ob->second->first + ob->second->count == i;
            if (joins) { b = ob->second; I.st.joined++; }
            else { b = std::make_shared<Impl::Batch>(); b->object = objectKey; b->first = i; }
            for (int c = i; c < j; c++) {
                auto pr = std::make_shared<std::promise<Bytes>>();
                b->slots.push_back(pr); b->count++;
                I.pending[{objectKey, c}] = pr->get_future().share();
            }
            if (i <= chunkId && chunkId < j) mine = I.pending[k];
            if (!joins) {
                if (i <= chunkId && chunkId < j && !lead) { lead = b; I.openBatch[objectKey] = b; }
                else {                                                  // pure prefetch run: sent from a helper thread, nobody waits here
                    b->open = false;
                    I.helpers.erase(std::remove_if(I.helpers.begin(), I.helpers.end(), [](std::future<void>& h) {
                                        return h.wait_for(std::chrono::seconds(0)) == std::future_status::ready; }), I.helpers.end());
                    const SegmentManifest copy = manifest;              // the helper may outlive the caller's object
                    I.helpers.push_back(std::async(std::launch::async, [this, b, copy] { impl_->runBatch(b, copy); }));
                }
            }
            i = j;
        }
        if (have) return cachedCopy;
    }
    if (lead) {
        // give concurrent misses on the next chunks of this object a moment to join (bounded, microseconds against get.timeout.ms)
        if (I.coalesceWaitMicros > 0) std::this_thread::sleep_for(std::chrono::microseconds(I.coalesceWaitMicros));
        { std::lock_guard<std::mutex> lk(I.mu); lead->open = false; auto ob = I.openBatch.find(objectKey); if (ob != I.openBatch.end() && ob->second == lead) I.openBatch.erase(ob); }
        I.runBatch(lead, manifest);
    }
    if (mine.wait_for(std::chrono::milliseconds(I.getTimeoutMs)) != std::future_status::ready)
        throw std::runtime_error("java.util.concurrent.TimeoutException");       // ChunkCache.java:126-128: wrapped in a RuntimeException
    return mine.get();
}

// =====================================================================================================
// FetchChunkEnumeration (fetch/FetchChunkEnumeration.java:53-138)
// =====================================================================================================
FetchChunkEnumeration::FetchChunkEnumeration(std::shared_ptr<ChunkManager> chunkManager, std::string objectKey, SegmentManifest manifest, BytesRange range)
    : chunkManager_(std::move(chunkManager)), objectKey_(std::move(objectKey)), manifest_(std::move(manifest)), range_(range) {
    if (!chunkManager_) throw std::invalid_argument("chunkManager cannot be null");
    if (!manifest_.chunkIndex) throw std::invalid_argument("manifest cannot be null");
    if (range_.to < range_.from) throw std::invalid_argument("range cannot be empty");
    const auto first = manifest_.chunkIndex->findChunkForOriginalOffset(range_.from);
    if (!first) throw std::invalid_argument("Invalid start position " + std::to_string(range_.from) + " in segment path " + objectKey_);
    startChunkId_ = currentChunkId_ = first->id;
    const auto last = manifest_.chunkIndex->findChunkForOriginalOffset(range_.to);
    lastChunkId_ = last ? last->id : manifest_.chunkIndex->chunks().back().id;
}

Bytes FetchChunkEnumeration::nextElement() {
    if (!hasMoreElements()) throw std::out_of_range("NoSuchElementException");
    Bytes content = chunkManager_->getChunk(objectKey_, manifest_, currentChunkId_);
    const Chunk& cur = manifest_.chunkIndex->chunks()[(size_t)currentChunkId_];
    const bool first = currentChunkId_ == startChunkId_, last = currentChunkId_ == lastChunkId_;
    size_t skip = first ? (size_t)(range_.from - cur.originalPosition) : 0;             // InputStream.skip
    if (skip > content.size()) skip = content.size();
    size_t end = content.size();
    if (last) {                                                                          // BoundedInputStream
        const size_t bound = first ? (size_t)range_.size() : (size_t)(range_.to - cur.originalPosition + 1);
        end = std::min(end, skip + bound);
    }
    currentChunkId_++;
    return Bytes(content.begin() + (long)skip, content.begin() + (long)end);
}

Bytes FetchChunkEnumeration::readAll() {
    Bytes out;
    while (hasMoreElements()) { const Bytes c = nextElement(); out.insert(out.end(), c.begin(), c.end()); }
    return out;
}

// =====================================================================================================
// upload sink
// =====================================================================================================
namespace {
class FinisherInputStream : public InputStream {        // new SequenceInputStream(transformFinisher)
public:
    explicit FinisherInputStream(TransformFinisher* f) : f_(f) {}
    long read(uint8_t* b, size_t len) override {
        while (pos_ >= cur_.size()) {
            if (!f_->hasMoreElements()) return -1;
            cur_ = f_->nextElement(); pos_ = 0;
        }
        const size_t m = std::min(len, cur_.size() - pos_);
        memcpy(b, cur_.data() + pos_, m);
        pos_ += m;
        return (long)m;
    }

private:
    TransformFinisher* f_;
    Bytes cur_;
    size_t pos_ = 0;
};
long long nowNs() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (long long)t.tv_sec * 1000000000ll + t.tv_nsec; }
}  // namespace

std::shared_ptr<InputStream> TransformFinisher::toInputStream(std::shared_ptr<TokenBucket> bucket) {
    std::shared_ptr<InputStream> s = std::make_shared<FinisherInputStream>(this);
    if (bucket) s = std::make_shared<RateLimitedInputStream>(s, std::move(bucket));
    return s;
}

TokenBucket::TokenBucket(int uploadRate) : rate_((double)std::max(uploadRate, MIN_RATE)), lastNs_(nowNs()) { tokens_ = rate_; }
void TokenBucket::refill() {
    const long long n = nowNs();
    tokens_ = std::min(rate_, tokens_ + (double)(n - lastNs_) * 1e-9 * rate_);
    lastNs_ = n;
}
void TokenBucket::consume(long tokens) {
    refill();
    tokens_ -= (double)tokens;                                     // may go negative: the debt is slept off, like a blocking consume
    if (tokens_ < 0) {
        const double wait = -tokens_ / rate_;
        timespec t; t.tv_sec = (time_t)wait; t.tv_nsec = (long)((wait - (double)t.tv_sec) * 1e9);
        nanosleep(&t, nullptr);
    }
}
void TokenBucket::forceAddTokens(long tokens) { tokens_ += (double)tokens; }

long RateLimitedInputStream::read(uint8_t* b, size_t len) {           // RateLimitedInputStream.java:56-84
    if (len > 0) bucket_->consume((long)len);
    const long r = in_->read(b, len);
    if (r > -1) { if ((long)len > r) bucket_->forceAddTokens((long)len - r); }
    else if (len > 0) bucket_->forceAddTokens((long)len);
    return r;
}

FileSystemStorage::FileSystemStorage(std::string root) : root_(std::move(root)) {
    struct stat st;
    if (stat(root_.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || access(root_.c_str(), W_OK) != 0) throw std::invalid_argument(root_ + " must be a writable directory");
}
long FileSystemStorage::upload(InputStream& in, const std::string& key) {
    const std::string path = root_ + "/" + key;
    for (size_t p = root_.size() + 1; (p = path.find('/', p)) != std::string::npos; p++) mkdir(path.substr(0, p).c_str(), 0777);   // Files.createDirectories
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("Failed to upload " + key);
    uint8_t buf[8192];                                               // InputStream.transferTo's buffer (Files.copy)
    long total = 0;
    for (;;) {
        const long r = in.read(buf, sizeof buf);
        if (r <= 0) break;
        if (fwrite(buf, 1, (size_t)r, f) != (size_t)r) { fclose(f); throw std::runtime_error("Failed to upload " + key); }
        total += r;
    }
    fclose(f);
    return total;
}
std::shared_ptr<InputStream> FileSystemStorage::fetch(const std::string& key, BytesRange range) {
    FILE* f = fopen((root_ + "/" + key).c_str(), "rb");
    if (!f) throw std::runtime_error("Key " + key + " does not exists in storage");          // KeyNotFoundException
    Bytes data((size_t)range.size());
    fseek(f, range.from, SEEK_SET);
    const size_t got = fread(data.data(), 1, data.size(), f);
    fclose(f);
    data.resize(got);
    return std::make_shared<ByteArrayInputStream>(std::move(data));
}

// =====================================================================================================
// neighbours of the path
// =====================================================================================================
bool segmentIsCompressed(Backend& be, const Bytes& seg) {
    // Kafka record batch v2 header: baseOffset 8 | batchLength 4 | partitionLeaderEpoch 4 | magic 1 | crc 4 | attributes 2 | ... (61 bytes)
    constexpr size_t LOG_OVERHEAD = 12, RECORD_BATCH_OVERHEAD = 61, MAGIC_OFFSET = 16, CRC_OFFSET = 17, ATTRIBUTES_OFFSET = 21;
    if (seg.size() < LOG_OVERHEAD + 5) throw InvalidRecordBatchException("Record batch is null");
    auto be32 = [&](size_t o) { return ((uint32_t)seg[o] << 24) | ((uint32_t)seg[o + 1] << 16) | ((uint32_t)seg[o + 2] << 8) | seg[o + 3]; };
    const uint32_t batchLength = be32(8);
    const size_t sizeInBytes = LOG_OVERHEAD + (size_t)batchLength;
    if (seg[MAGIC_OFFSET] != 2) throw InvalidRecordBatchException("Failed to read and validate first batch: unsupported magic " + std::to_string(seg[MAGIC_OFFSET]));
    if (sizeInBytes < RECORD_BATCH_OVERHEAD)
        throw InvalidRecordBatchException("Record batch is corrupt (the size " + std::to_string(sizeInBytes) + " is smaller than the minimum allowed overhead " +
                                          std::to_string(RECORD_BATCH_OVERHEAD) + ")");
    if (sizeInBytes > seg.size()) throw InvalidRecordBatchException("Record batch is null");                    // FileRecords.firstBatch(): incomplete batch
    const uint32_t stored = be32(CRC_OFFSET);
    const uint32_t computed = be.crc32c(seg.data() + ATTRIBUTES_OFFSET, sizeInBytes - ATTRIBUTES_OFFSET);
    if (stored != computed)
        throw InvalidRecordBatchException("Record is corrupt (stored crc = " + std::to_string(stored) + ", computed crc = " + std::to_string(computed) + ")");
    const uint32_t attributes = ((uint32_t)seg[ATTRIBUTES_OFFSET] << 8) | seg[ATTRIBUTES_OFFSET + 1];
    return (attributes & 0x07) != 0;                                                                             // CompressionType.NONE has id 0
}

Bytes transformIndex(std::shared_ptr<Backend> be, const Bytes& index, const std::optional<DataKeyAndAAD>& key, IvSupplier iv) {
    if (index.empty()) return {};
    std::shared_ptr<TransformChunkEnumeration> e = std::make_shared<BaseTransformChunkEnumeration>(std::make_shared<ByteArrayInputStream>(index), (int)index.size());
    if (key) e = std::make_shared<GpuTransformChunkEnumeration>(be, e, false, key, std::move(iv), 1);
    TransformFinisher f(e, (int)index.size(), false);                 // withChunkingDisabled()
    Bytes out = f.nextElement();
    const auto chunkIndex = f.chunkIndex();
    if (chunkIndex->chunks().size() != 1) throw std::logic_error("Number of chunks different than 1, single chunk is expected");
    if ((size_t)chunkIndex->chunks()[0].range().size() != out.size()) throw std::logic_error("chunk index and transformed index disagree");
    return out;
}

}  // namespace tsx
