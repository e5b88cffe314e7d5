// Host-side tests of the chunk-transform path (tsxhost over libtsxform's C ABI), written to read like the reference's own
// JUnit tests — each TEST names the reference test it restates (CT/ = core/src/test/java/io/aiven/kafka/tieredstorage/).
// Run by tests/test_host.py:  host_tests cpu            (pure host logic, no device library)
//                             host_tests backend <lib>  (chain through a libtsxform build: the emulated one on CPU, the real one -m gpu)
// The oracle (oracle/_build/liboracle.so) is linked here as the checker only.
#include <time.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <atomic>
#include <random>
#include <thread>
#include <string>

#include "../../tiered-storage-for-apache-kafka_amd/host/tsxhost.hpp"

extern "C" {
uint32_t orc_crc32c(const uint8_t* p, size_t n);
long orc_gcm_decrypt_chunk(const uint8_t key[32], const uint8_t* aad, size_t aad_len, const uint8_t* chunk, size_t len, uint8_t* out);
size_t orc_transform_chunk(unsigned flags, const uint8_t key[32], const uint8_t* aad, size_t aad_len, const uint8_t iv[12], const uint8_t* src, size_t n,
                           uint8_t* dst, size_t cap, uint8_t* scratch, uint32_t* crc_out);
size_t orc_chain_bound(size_t n, unsigned flags);
const char* orc_zstd_version(void);
}

using namespace tsx;
static int g_failed = 0, g_run = 0;
static std::string g_lib;
#define CHECK(c) do { if (!(c)) { printf("    FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); throw std::runtime_error("check failed"); } } while (0)
template <class E, class F> static void expectThrows(F f, const char* msg) {
    try { f(); } catch (const E& e) { if (msg && std::string(e.what()) != msg) { printf("    wrong message: '%s' (expected '%s')\n", e.what(), msg); throw std::runtime_error("check failed"); } return; }
    printf("    expected exception '%s' was not thrown\n", msg ? msg : "");
    throw std::runtime_error("check failed");
}
static void run(const char* name, const std::function<void()>& f) {
    g_run++;
    try { f(); printf("  ok   %s\n", name); } catch (const std::exception& e) { g_failed++; printf("  FAIL %s: %s\n", name, e.what()); }
}
static Bytes randomBytes(size_t n, uint32_t seed) { std::mt19937 r(seed); Bytes b(n); for (auto& x : b) x = (uint8_t)r(); return b; }
static Bytes textBytes(size_t n, uint32_t seed) {       // compressible, record-like
    std::mt19937 r(seed); Bytes b; b.reserve(n + 128); int id = 0;
    while (b.size() < n) {
        char rec[160];
        int len = snprintf(rec, sizeof rec, "{\"id\":%010d,\"user\":\"u%05u\",\"event\":\"%s\",\"payload\":\"", id++, (unsigned)(r() % 1000), (r() & 1) ? "click" : "views");
        b.insert(b.end(), rec, rec + len);
        for (unsigned k = 8 + r() % 40; k; k--) b.push_back((uint8_t)('a' + r() % 26));
        b.push_back('"'); b.push_back('}'); b.push_back('\n');
    }
    b.resize(n);
    return b;
}
static std::shared_ptr<InputStream> stream(const Bytes& b) { return std::make_shared<ByteArrayInputStream>(b); }

// ---------------------------------------------------------------------------------------------------------
static void cpuTests() {
    // CT/transform/BaseTransformChunkEnumerationTest.java:45-94
    run("BaseTransformChunkEnumerationTest.negativeChunkSize", [] {
        expectThrows<std::invalid_argument>([] { BaseTransformChunkEnumeration e(stream({}), -1); }, "originalChunkSize must be non-negative, -1 given");
    });
    run("BaseTransformChunkEnumerationTest.emptyInputStream", [] {
        BaseTransformChunkEnumeration e(stream({}), 1);
        CHECK(!e.hasMoreElements());
        expectThrows<std::out_of_range>([&] { e.nextElement(); }, nullptr);
    });
    run("BaseTransformChunkEnumerationTest.inAllCases (10 bytes @3 -> 3,3,3,1; chunk > file; chunk == 0)", [] {
        const Bytes data = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
        BaseTransformChunkEnumeration e(stream(data), 3);
        CHECK(e.originalChunkSize() == 3 && *e.transformedChunkSize() == 3);
        std::vector<size_t> sizes; Bytes all;
        while (e.hasMoreElements()) { Bytes c = e.nextElement(); sizes.push_back(c.size()); all.insert(all.end(), c.begin(), c.end()); }
        CHECK((sizes == std::vector<size_t>{3, 3, 3, 1}) && all == data);
        BaseTransformChunkEnumeration big(stream(data), 100);
        CHECK(big.nextElement() == data && !big.hasMoreElements());
        BaseTransformChunkEnumeration none(stream(data), 0);           // chunking disabled: readAllBytes
        CHECK(none.nextElement() == data && !none.hasMoreElements());
    });
    // CT/manifest/index/ChunkIndexBuilderCommonTest.java:37-127 for both builders
    for (int kind = 0; kind < 2; kind++) {
        auto normal = [kind]() -> std::unique_ptr<AbstractChunkIndexBuilder> {
            if (kind == 0) return std::make_unique<FixedSizeChunkIndexBuilder>(100, 250, 113);
            return std::make_unique<VariableSizeChunkIndexBuilder>(100, 250);
        };
        auto emptyFile = [kind]() -> std::unique_ptr<AbstractChunkIndexBuilder> {
            if (kind == 0) return std::make_unique<FixedSizeChunkIndexBuilder>(100, 0, 113);
            return std::make_unique<VariableSizeChunkIndexBuilder>(100, 0);
        };
        const std::string pre = kind == 0 ? "FixedSizeChunkIndexBuilderTest." : "VariableSizeChunkIndexBuilderTest.";
        run((pre + "addChunkToAlreadyFinishedIndex").c_str(), [&] {
            auto b = normal(); b->addChunk(113); b->addChunk(113); b->finish(113);
            expectThrows<std::logic_error>([&] { b->addChunk(113); }, "Cannot add chunk to already finished index");
        });
        run((pre + "addNonPositiveSizedChunk").c_str(), [&] {
            auto b = normal();
            expectThrows<std::invalid_argument>([&] { b->addChunk(-1); }, "Transformed chunk size must be non-negative, -1 given");
        });
        run((pre + "addSupposedToBeFinalChunkAsNonFinal").c_str(), [&] {
            auto b = normal(); b->addChunk(113); b->addChunk(113);
            expectThrows<std::logic_error>([&] { b->addChunk(113); }, "This must be final chunk. Call `finish` instead.");
        });
        run((pre + "finishedAlreadyFinishedIndex").c_str(), [&] {
            auto b = normal(); b->addChunk(113); b->addChunk(113); b->finish(113);
            expectThrows<std::logic_error>([&] { b->finish(113); }, "Cannot finish already finished index");
        });
        run((pre + "finishWithNegativeSizedChunk").c_str(), [&] {
            auto b = normal(); b->addChunk(113); b->addChunk(113);
            expectThrows<std::invalid_argument>([&] { b->finish(-1); }, "Transformed chunk size must be non-negative, -1 given");
        });
        run((pre + "finishWithNonFinalChunk").c_str(), [&] {
            auto b = normal(); b->addChunk(113);
            expectThrows<std::logic_error>([&] { b->finish(113); }, "This cannot be final chunk: not enough chunks to cover original file. Call `addChunk` instead.");
        });
        run((pre + "findForNegativeOffset / findForEmptyFile / findBeyondFileBorder").c_str(), [&] {
            auto b = normal(); b->addChunk(113); b->addChunk(113); auto index = b->finish(113);
            expectThrows<std::invalid_argument>([&] { index->findChunkForOriginalOffset(-1); }, "Offset must be non-negative, -1 given");
            CHECK(!index->findChunkForOriginalOffset(250) && !index->findChunkForOriginalOffset(251));
            auto e = emptyFile()->finish(0);
            CHECK(!e->findChunkForOriginalOffset(0) && !e->findChunkForOriginalOffset(1));
            CHECK(e->chunks().size() == 1 && e->chunks()[0] == (Chunk{0, 0, 0, 0, 0}));
        });
    }
    // CT/manifest/index/FixedSizeChunkIndexBuilderTest.java:48-88
    run("FixedSizeChunkIndexBuilderTest.addInvalidNonFinalTransformedChunkSize", [] {
        FixedSizeChunkIndexBuilder b(100, 250, 113);
        expectThrows<std::invalid_argument>([&] { b.addChunk(12); }, "Non-final chunk must be of size 113, but 12 given");
    });
    run("FixedSizeChunkIndexBuilderTest.threeChunks", [] {
        FixedSizeChunkIndexBuilder b(101, 253, 111); b.addChunk(111); b.addChunk(111); auto index = b.finish(81);
        const Chunk c1{0, 0, 101, 0, 111}, c2{1, 101, 101, 111, 111}, c3{2, 202, 51, 222, 81};
        CHECK((index->chunks() == std::vector<Chunk>{c1, c2, c3}));
        for (int i = 0; i < 101; i++) CHECK(*index->findChunkForOriginalOffset(i) == c1);
        for (int i = 101; i < 202; i++) CHECK(*index->findChunkForOriginalOffset(i) == c2);
        for (int i = 202; i < 253; i++) CHECK(*index->findChunkForOriginalOffset(i) == c3);
        CHECK(!index->findChunkForOriginalOffset(253) && !index->findChunkForOriginalOffset(254));
    });
    // CT/manifest/index/VariableSizeChunkIndexBuilderTest.java:39-81
    run("VariableSizeChunkIndexBuilderTest.invalidSizes + threeChunks", [] {
        expectThrows<std::invalid_argument>([] { VariableSizeChunkIndexBuilder b(100, -1); }, "Original file size must be non-negative, -1 given");
        VariableSizeChunkIndexBuilder b(101, 253); b.addChunk(33); b.addChunk(22); auto index = b.finish(5);
        const Chunk c1{0, 0, 101, 0, 33}, c2{1, 101, 101, 33, 22}, c3{2, 202, 51, 55, 5};
        CHECK((index->chunks() == std::vector<Chunk>{c1, c2, c3}));
        for (int i = 0; i < 253; i++) CHECK(*index->findChunkForOriginalOffset(i) == (i < 101 ? c1 : i < 202 ? c2 : c3));
        CHECK(!index->findChunkForOriginalOffset(253));
    });
    // CT/manifest/index/serde/ChunkSizesBinaryCodecTest.java:36-113
    run("ChunkSizesBinaryCodecTest.empty / singleValue / multipleValues / negativeValues", [] {
        CHECK(ChunkSizesBinaryCodec::encode({}).size() == 4 && ChunkSizesBinaryCodec::decode(ChunkSizesBinaryCodec::encode({})).empty());
        for (int v : {0, 1, 0x7FFFFFFF}) { const Bytes e = ChunkSizesBinaryCodec::encode({v}); CHECK(e.size() == 8 && ChunkSizesBinaryCodec::decode(e) == std::vector<int>{v}); }
        const int MAX = 0x7FFFFFFF;
        struct Case { std::vector<int> v; int bpv; };
        std::vector<int> longList; for (long long i = 0; i < (long long)MAX - 2000; i += 1000) longList.push_back((int)i);
        std::vector<int> rev(longList.rbegin(), longList.rend());
        const std::vector<Case> cases = {{{0, 1000, 2, 44002, 369}, 2}, {{MAX, MAX - 1, MAX - 2, 10}, 1}, {{MAX / 2, MAX / 2 - 1, MAX / 2 - 2, 10}, 1}, {longList, 4}, {rev, 4},
                                         {{1, 2, 3, MAX}, 1}, {{1, 0xFF + 10, 0xFF + 20, 0xFF + 30, MAX}, 2}, {{1, 0xFFFF + 10, 0xFFFF + 20, 0xFFFF + 30, MAX}, 3},
                                         {{1, 0xFFFFFF + 10, 0xFFFFFF + 20, 0xFFFFFF + 30, MAX}, 4}};
        for (const auto& c : cases) {
            const Bytes e = ChunkSizesBinaryCodec::encode(c.v);
            CHECK(e.size() == 4 + 4 + 1 + (c.v.size() - 1) * (size_t)c.bpv + 4);
            CHECK((((uint32_t)e[0] << 24) | ((uint32_t)e[1] << 16) | ((uint32_t)e[2] << 8) | e[3]) == c.v.size() && e[8] == c.bpv);
            CHECK(ChunkSizesBinaryCodec::decode(e) == c.v);
        }
        expectThrows<std::invalid_argument>([] { ChunkSizesBinaryCodec::encode({-1}); }, "Values cannot be negative");
        expectThrows<std::invalid_argument>([] { ChunkSizesBinaryCodec::encode({1, -1, 1}); }, "Values cannot be negative");
        expectThrows<std::invalid_argument>([] { ChunkSizesBinaryCodec::encode({1, 1, -1}); }, "Values cannot be negative");
    });
    // CT/transform/TransformFinisherTest.java:46-125 (the parts that need no transform)
    run("TransformFinisherTest.getIndexBeforeUsing / nullInnerEnumeration / negativeOriginalFileSize", [] {
        struct Fake : TransformChunkEnumeration {              // a non-base enumeration with a fixed transformed size
            std::shared_ptr<BaseTransformChunkEnumeration> b; Fake(std::shared_ptr<BaseTransformChunkEnumeration> x) : b(std::move(x)) {}
            int originalChunkSize() const override { return b->originalChunkSize(); }
            std::optional<int> transformedChunkSize() const override { return b->transformedChunkSize(); }
            bool hasMoreElements() override { return b->hasMoreElements(); }
            Bytes nextElement() override { return b->nextElement(); }
        };
        auto base = std::make_shared<BaseTransformChunkEnumeration>(stream({1, 2, 3, 4, 5, 6, 7}), 3);
        TransformFinisher f(std::make_shared<Fake>(base), 7);
        expectThrows<std::logic_error>([&] { f.chunkIndex(); }, "Chunk index was not built, was finisher used?");
        expectThrows<std::invalid_argument>([] { TransformFinisher f2(nullptr, 7); }, "inner cannot be null");
        expectThrows<std::invalid_argument>([&] { TransformFinisher f3(base, -1); }, "originalFileSize must be non-negative, -1 given");
    });
    run("TransformFinisherTest.buildIndexWhenInnerEnumerationHasFixedSize (7 bytes @3, base transform)", [] {
        auto base = std::make_shared<BaseTransformChunkEnumeration>(stream({1, 2, 3, 4, 5, 6, 7}), 3);
        TransformFinisher f(base, 7);
        const Bytes all = f.toBytes();
        CHECK(all == (Bytes{1, 2, 3, 4, 5, 6, 7}));
        auto idx = f.chunkIndex();
        CHECK(idx->isFixed());
        CHECK((idx->chunks() == std::vector<Chunk>{{0, 0, 3, 0, 3}, {1, 3, 3, 3, 3}, {2, 6, 1, 6, 1}}));
        // the arithmetic path (index asked without consuming a base transform, TransformFinisher.java:124-132)
        TransformFinisher g(std::make_shared<BaseTransformChunkEnumeration>(stream({1, 2, 3, 4, 5, 6, 7}), 3), 7);
        CHECK(g.chunkIndex()->chunks() == idx->chunks());
    });
    // CT/transform/BaseDetransformChunkEnumerationTest.java:47-117
    run("BaseDetransformChunkEnumerationTest.*", [] {
        BaseDetransformChunkEnumeration empty(stream({}), {Chunk{0, 0, 0, 0, 0}});
        CHECK(!empty.hasMoreElements());
        const Bytes data = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
        auto in = std::make_shared<ByteArrayInputStream>(data);
        BaseDetransformChunkEnumeration e(in, {Chunk{0, 0, 3, 0, 3}, Chunk{1, 3, 3, 3, 3}, Chunk{2, 6, 3, 6, 3}});       // extra bytes are ignored
        CHECK(e.nextElement() == (Bytes{0, 1, 2}) && e.nextElement() == (Bytes{3, 4, 5}) && e.nextElement() == (Bytes{6, 7, 8}));
        CHECK(!e.hasMoreElements() && in->closed());
        expectThrows<std::out_of_range>([&] { e.nextElement(); }, nullptr);
        BaseDetransformChunkEnumeration shortStream(stream({0, 1, 2, 3}), {Chunk{0, 0, 3, 0, 3}, Chunk{1, 3, 3, 3, 3}});
        CHECK(shortStream.nextElement() == (Bytes{0, 1, 2}));
        expectThrows<std::runtime_error>([&] { shortStream.hasMoreElements(); }, "Stream has fewer bytes than expected");
        BaseDetransformChunkEnumeration noChunks(stream(data));                // empty chunk list: everything at once
        CHECK(noChunks.nextElement() == data && !noChunks.hasMoreElements());
        BaseDetransformChunkEnumeration raw(stream(data));                     // DetransformFinisherTest.java:28-49: base -> the raw stream
        CHECK(DetransformFinisher(std::make_shared<BaseDetransformChunkEnumeration>(stream(data))).toBytes() == data);
    });
    run("base64 (java.util.Base64 basic alphabet, padding)", [] {
        CHECK(base64Encode({}) == "" && base64Encode({'f'}) == "Zg==" && base64Encode({'f', 'o'}) == "Zm8=" && base64Encode({'f', 'o', 'o', 'b', 'a', 'r'}) == "Zm9vYmFy");
        CHECK(base64Decode("Zm9vYmE=") == (Bytes{'f', 'o', 'o', 'b', 'a'}));
    });
}

// ---------------------------------------------------------------------------------------------------------
static const Bytes KEY = [] { Bytes k(32); for (int i = 0; i < 32; i++) k[(size_t)i] = (uint8_t)i; return k; }();
static const Bytes AAD = [] { Bytes k(32); for (int i = 0; i < 32; i++) k[(size_t)i] = (uint8_t)(32 + i); return k; }();
static IvSupplier countingIv() { auto n = std::make_shared<uint64_t>(0); return [n](uint8_t iv[12]) { memset(iv, 0, 12); uint64_t v = (*n)++; for (int i = 0; i < 8; i++) iv[11 - i] = (uint8_t)(v >> (8 * i)); }; }

struct MemFetcher : ObjectFetcher {
    Bytes object; std::atomic<int> fetches{0};
    std::shared_ptr<InputStream> fetch(const std::string&, BytesRange r) override {
        fetches++;
        return std::make_shared<ByteArrayInputStream>(Bytes(object.begin() + r.from, object.begin() + r.to + 1));
    }
};

struct StubChunkManager : ChunkManager {                  // every chunk is "0123456789" (FetchChunkEnumerationTest.java:58)
    std::vector<int> asked;
    Bytes getChunk(const std::string&, const SegmentManifest&, int chunkId) override {
        asked.push_back(chunkId);
        const char* c = "0123456789";
        return Bytes(c, c + 10);
    }
};

static void segmentIndexesBuilderTests() {
    // CT/manifest/SegmentIndexesV1BuilderTest.java:30-95
    run("SegmentIndexesV1BuilderTest: failures and both layouts", [] {
        expectThrows<std::logic_error>([] { SegmentIndexesV1Builder().build(); }, "Not enough indexes have been added; at least 4 required. Indexes included: []");
        expectThrows<std::logic_error>([] { SegmentIndexesV1Builder().add(IndexType::OFFSET, 1).add(IndexType::TIMESTAMP, 1).build(); },
                                       "Not enough indexes have been added; at least 4 required. Indexes included: [OFFSET, TIMESTAMP]");
        expectThrows<std::logic_error>([] { SegmentIndexesV1Builder().add(IndexType::OFFSET, 1).add(IndexType::OFFSET, 1); }, "Index OFFSET is already added");
        expectThrows<std::logic_error>([] { SegmentIndexesV1Builder().add(IndexType::OFFSET, 1).add(IndexType::TIMESTAMP, 1).add(IndexType::PRODUCER_SNAPSHOT, 1)
                                                .add(IndexType::TRANSACTION, 1).build(); }, "OFFSET, TIMESTAMP, PRODUCER_SNAPSHOT, and LEADER_EPOCH indexes are required");
        const SegmentIndexesV1 a = SegmentIndexesV1Builder().add(IndexType::OFFSET, 1).add(IndexType::TIMESTAMP, 1).add(IndexType::PRODUCER_SNAPSHOT, 1)
                                       .add(IndexType::LEADER_EPOCH, 1).add(IndexType::TRANSACTION, 1).build();
        CHECK(a.offset == (SegmentIndexV1{0, 1}) && a.timestamp == (SegmentIndexV1{1, 1}) && a.producerSnapshot == (SegmentIndexV1{2, 1}) &&
              a.leaderEpoch == (SegmentIndexV1{3, 1}) && a.transaction && *a.transaction == (SegmentIndexV1{4, 1}));
        const SegmentIndexesV1 b = SegmentIndexesV1Builder().add(IndexType::OFFSET, 10).add(IndexType::TIMESTAMP, 20).add(IndexType::PRODUCER_SNAPSHOT, 0)
                                       .add(IndexType::LEADER_EPOCH, 7).build();
        CHECK(b.timestamp == (SegmentIndexV1{10, 20}) && b.producerSnapshot == (SegmentIndexV1{30, 0}) && b.leaderEpoch == (SegmentIndexV1{30, 7}) && !b.transaction);
    });
}

static void fetchEnumerationTests() {
    // CT/fetch/FetchChunkEnumerationTest.java:46-160: 10 chunks of 10 bytes (fixed index 10 / 100 / 12 / 12)
    SegmentManifest m; m.chunkIndex = std::make_shared<FixedSizeChunkIndex>(10, 100, 12, 12);
    auto str = [](const Bytes& b) { return std::string(b.begin(), b.end()); };
    run("FetchChunkEnumerationTest.failsWhenLargerStartPosition", [&] {
        auto cm = std::make_shared<StubChunkManager>();
        expectThrows<std::invalid_argument>([&] { FetchChunkEnumeration e(cm, "topic/segment", m, BytesRange{1000, 1001}); }, "Invalid start position 1000 in segment path topic/segment");
    });
    run("FetchChunkEnumerationTest.endPositionIsWithinIndex / endPositionIsOutsideIndex", [&] {
        auto cm = std::make_shared<StubChunkManager>();
        FetchChunkEnumeration a(cm, "topic/segment", m, BytesRange{0, 80});
        CHECK(a.startChunkId() == 0 && a.lastChunkId() == 8);
        FetchChunkEnumeration b(cm, "topic/segment", m, BytesRange{0, 110});
        CHECK(b.startChunkId() == 0 && b.lastChunkId() == 9);
        CHECK(cm->asked.empty());                                                        // construction fetches nothing
    });
    run("FetchChunkEnumerationTest.shouldReturnRangeFromSingleChunk", [&] {
        auto cm = std::make_shared<StubChunkManager>();
        FetchChunkEnumeration e(cm, "topic/segment", m, BytesRange{32, 34});
        CHECK(e.startChunkId() == e.lastChunkId());
        CHECK(str(e.nextElement()) == "234");
        CHECK(!e.hasMoreElements());
        expectThrows<std::out_of_range>([&] { e.nextElement(); }, nullptr);
    });
    run("FetchChunkEnumerationTest.shouldReturnRangeFromMultipleChunks + laziness (close early asks for nothing more)", [&] {
        auto cm = std::make_shared<StubChunkManager>();
        FetchChunkEnumeration e(cm, "topic/segment", m, BytesRange{15, 34});
        CHECK(e.startChunkId() != e.lastChunkId());
        CHECK(str(e.nextElement()) == "56789");
        CHECK((cm->asked == std::vector<int>{1}));                                       // one chunk read, one chunk asked for
        CHECK(str(e.nextElement()) == "0123456789");
        CHECK(str(e.nextElement()) == "01234");
        CHECK(!e.hasMoreElements());
        expectThrows<std::out_of_range>([&] { e.nextElement(); }, nullptr);
        auto cm2 = std::make_shared<StubChunkManager>();
        FetchChunkEnumeration lazy(cm2, "topic/segment", m, BytesRange{5, 95});
        CHECK(str(lazy.nextElement()) == "56789");
        lazy.close();
        CHECK(!lazy.hasMoreElements() && (cm2->asked == std::vector<int>{0}));
        FetchChunkEnumeration all(std::make_shared<StubChunkManager>(), "topic/segment", m, BytesRange{7, 1000});
        CHECK(all.readAll().size() == 93);
    });
}

static void backendTests(bool full) {
    auto be = std::make_shared<Backend>(g_lib);
    printf("  backend: %s\n", be->version().c_str());
    const int ORIGINAL_SIZE = full ? 1812004 : 70000;
    const Bytes original = randomBytes((size_t)ORIGINAL_SIZE, 1), text = textBytes((size_t)ORIGINAL_SIZE, 2);

    // CT/transform/TransformsEndToEndTest.java:44-108 — same chunk sizes, all four chains, transform then detransform
    auto endToEnd = [&](const Bytes& data, int chunkSize, bool compression, bool encryption) {
        std::shared_ptr<TransformChunkEnumeration> t = std::make_shared<BaseTransformChunkEnumeration>(stream(data), chunkSize);
        if (compression || encryption)
            t = std::make_shared<GpuTransformChunkEnumeration>(be, t, compression, encryption ? std::optional<DataKeyAndAAD>(DataKeyAndAAD{KEY, AAD}) : std::nullopt, countingIv(), 16);
        TransformFinisher tf(t, (int)data.size(), chunkSize != 0);
        const Bytes uploaded = tf.toBytes();
        auto index = tf.chunkIndex();
        CHECK(index->isFixed() == !compression);                                     // TransformFinisherTest.java:97-125: variable iff transformedChunkSize() is null
        if (encryption && !compression && chunkSize != 0) CHECK(index->chunks()[0].transformedSize == std::min(chunkSize, (int)data.size()) + 28);
        std::shared_ptr<DetransformChunkEnumeration> d = std::make_shared<BaseDetransformChunkEnumeration>(stream(uploaded), index->chunks());
        if (compression || encryption)
            d = std::make_shared<GpuDetransformChunkEnumeration>(be, d, compression, encryption ? std::optional<SegmentEncryptionMetadata>(SegmentEncryptionMetadata{KEY, AAD, IV_SIZE}) : std::nullopt,
                                                                 chunkSize == 0 ? (int)data.size() : chunkSize, 16);
        CHECK(DetransformFinisher(d).toBytes() == data);
    };
    // the object assembled in place (TSX_MEM_HOST_PACKED through TransformFinisher::toBytesPacked) is byte for
    // byte the object the per-chunk path builds, with the same chunk index - for every chain and for chunking disabled
    auto packedEqualsChunked = [&](const Bytes& data, int chunkSize, bool compression, bool encryption) {
        auto make = [&] {
            std::shared_ptr<TransformChunkEnumeration> t = std::make_shared<BaseTransformChunkEnumeration>(stream(data), chunkSize);
            if (compression || encryption)
                t = std::make_shared<GpuTransformChunkEnumeration>(be, t, compression, encryption ? std::optional<DataKeyAndAAD>(DataKeyAndAAD{KEY, AAD}) : std::nullopt, countingIv(), 7);
            return t;
        };
        TransformFinisher a(make(), (int)data.size(), chunkSize != 0), b(make(), (int)data.size(), chunkSize != 0);
        const Bytes chunked = a.toBytes(), packed = b.toBytesPacked();
        CHECK(packed == chunked);
        auto ia = a.chunkIndex(), ib = b.chunkIndex();
        CHECK(ia->isFixed() == ib->isFixed() && ia->chunks().size() == ib->chunks().size());
        for (size_t i = 0; i < ia->chunks().size(); i++)
            CHECK(ia->chunks()[i].transformedPosition == ib->chunks()[i].transformedPosition && ia->chunks()[i].transformedSize == ib->chunks()[i].transformedSize);
    };
    run("TransformFinisher.toBytesPacked (upload sink without per-chunk copies)", [&] {
        for (int c : {0, 4096 + 3, 16384})
            for (int mode = 1; mode < 4; mode++) packedEqualsChunked(text, c, (mode & 1) != 0, (mode & 2) != 0);
    });
    // SURVEY §8 f3, the JVM half (java/.../GpuTransformFinisher.java; twin tsx::GpuTransformFinisher): the uploader reads the object out of
    // packed batch buffers - no array per chunk, no SequenceInputStream - and gets the bytes and the chunk index of the reference's path
    // (TransformFinisher.java:134-151: SequenceInputStream(this), optionally rate limited), for every chain, with and without read-ahead,
    // through the stream and through a part-buffer sink (S3MultiPartOutputStream.java:89-122)
    run("GpuTransformFinisher: object + index equal the SequenceInputStream path, part sink, index only after the drain", [&] {
        auto sameIndex = [&](const std::shared_ptr<ChunkIndex>& ia, const std::shared_ptr<ChunkIndex>& ib) {
            CHECK(ia->isFixed() == ib->isFixed() && ia->chunks().size() == ib->chunks().size());
            for (size_t i = 0; i < ia->chunks().size() && i < ib->chunks().size(); i++)
                CHECK(ia->chunks()[i].transformedPosition == ib->chunks()[i].transformedPosition && ia->chunks()[i].transformedSize == ib->chunks()[i].transformedSize &&
                      ia->chunks()[i].originalPosition == ib->chunks()[i].originalPosition && ia->chunks()[i].originalSize == ib->chunks()[i].originalSize);
        };
        for (int c : {0, 4096 + 3, 16384})
            for (int mode = 1; mode < 4; mode++)
                for (int ra = 0; ra < 2; ra++) {
                    const bool compression = (mode & 1) != 0, encryption = (mode & 2) != 0;
                    auto make = [&](int batch) {
                        std::shared_ptr<TransformChunkEnumeration> base = std::make_shared<BaseTransformChunkEnumeration>(stream(text), c);
                        return std::make_shared<GpuTransformChunkEnumeration>(be, base, compression, encryption ? std::optional<DataKeyAndAAD>(DataKeyAndAAD{KEY, AAD}) : std::nullopt,
                                                                              countingIv(), batch, false, TSX_ZSTD_PROFILE_1_5_7, ra != 0);
                    };
                    TransformFinisher a(make(7), (int)text.size(), c != 0);
                    const Bytes chunked = a.toInputStream()->readAllBytes();
                    GpuTransformFinisher b(make(7), (int)text.size(), c != 0, nullptr, ra != 0);
                    try { b.chunkIndex(); CHECK(false); } catch (const std::logic_error& e) { CHECK(std::string(e.what()) == "Chunk index was not built, was finisher used?"); }
                    const Bytes packed = b.toInputStream()->readAllBytes();
                    CHECK(packed == chunked);
                    sameIndex(a.chunkIndex(), b.chunkIndex());
                    GpuTransformFinisher p(make(5), (int)text.size(), c != 0, nullptr, ra != 0);
                    Bytes parts; std::vector<uint8_t> part(5000);
                    for (;;) { const size_t m = p.fillPart(part.data(), part.size()); parts.insert(parts.end(), part.begin(), part.begin() + (long)m); if (m < part.size()) break; }
                    CHECK(parts == chunked);
                    sameIndex(a.chunkIndex(), p.chunkIndex());
                }
    });
    run("GpuTransformFinisher: the rate limit is honoured (RateLimitedInputStream.java:56-84 around the packed stream)", [&] {
        const Bytes data = randomBytes(40 * 1024 - 3 * 28, 5);            // three encrypted chunks: 40 KiB on the wire
        auto make = [&] {
            std::shared_ptr<TransformChunkEnumeration> base = std::make_shared<BaseTransformChunkEnumeration>(stream(data), 16 * 1024);
            return std::make_shared<GpuTransformChunkEnumeration>(be, base, false, DataKeyAndAAD{KEY, AAD}, countingIv(), 2);
        };
        TransformFinisher ref(make(), (int)data.size());
        const Bytes expect = ref.toInputStream()->readAllBytes();
        GpuTransformFinisher f(make(), (int)data.size(), true, std::make_shared<TokenBucket>(16384));
        timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
        const Bytes got = f.toInputStream()->readAllBytes();              // 40 KiB at 16 KiB/s from a full 16 KiB bucket: >= 1.4 s
        clock_gettime(CLOCK_MONOTONIC, &b);
        const double el = (double)(b.tv_sec - a.tv_sec) + (double)(b.tv_nsec - a.tv_nsec) * 1e-9;
        CHECK(got == expect && got.size() == 40 * 1024);
        CHECK(el > 1.2 && el < 3.5);
        CHECK(f.chunkIndex()->chunks().size() == 3);
    });
    const int S = ORIGINAL_SIZE;
    run("TransformsEndToEndTest.plaintext", [&] { for (int c : {0, 1024, 1024 * 2, 1024 * 5 + 3, S - 1, S * 2}) endToEnd(original, c, false, false); });
    run("TransformsEndToEndTest.encryption", [&] { for (int c : {0, 1024 * 5 + 3, 16384 + 2, S - 1, S * 2}) endToEnd(original, c, false, true); });
    run("TransformsEndToEndTest.compression", [&] { for (int c : {1024 * 5 + 3, 16384, S - 1, S * 2}) { endToEnd(original, c, true, false); endToEnd(text, c, true, false); } });
    run("TransformsEndToEndTest.compressionAndEncryption", [&] { for (int c : {1024 * 5 + 3, 16384, S - 1, S * 2}) { endToEnd(original, c, true, true); endToEnd(text, c, true, true); } });

    // CT/transform/EncryptionChunkEnumerationTest.java:79-110, DecryptionChunkEnumerationTest.java:81-95
    run("EncryptionChunkEnumerationTest.transformedChunkSizePropagation + layout IV||C||TAG", [&] {
        auto base = std::make_shared<BaseTransformChunkEnumeration>(stream(text), 4096);
        GpuTransformChunkEnumeration enc(be, base, false, DataKeyAndAAD{KEY, AAD}, countingIv(), 4);
        CHECK(enc.originalChunkSize() == 4096 && *enc.transformedChunkSize() == 4096 + 12 + 16);
        GpuTransformChunkEnumeration comp(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), 4096), true, DataKeyAndAAD{KEY, AAD}, countingIv(), 4);
        CHECK(!comp.transformedChunkSize());                                                         // CompressionChunkEnumeration.java:39-42
        const Bytes c0 = enc.nextElement(), c1 = enc.nextElement();
        CHECK(c0.size() == 4096 + 28 && c0[11] == 0 && c1[11] == 1);                                // the IVs of the supplier, in order, in front
        Bytes exp(orc_chain_bound(4096, 2)); uint32_t crc; uint8_t iv0[12] = {0};
        Bytes scratch(orc_chain_bound(4096, 1) + 64);
        const size_t m = orc_transform_chunk(2 /*encrypt*/, KEY.data(), AAD.data(), AAD.size(), iv0, text.data(), 4096, exp.data(), exp.size(), scratch.data(), &crc);
        exp.resize(m);
        CHECK(c0 == exp);                                                                            // ciphertext + tag bit-exact vs the oracle
    });
    // read-ahead: batch k + 1 is on the device while the consumer drains batch k - same chunks, same IV order, same CRC list, the
    // same chunk index; a failure in batch k + 1 surfaces only after batch k has been handed out; an enumeration dropped with its
    // helper still running goes away cleanly
    run("GpuTransformChunkEnumeration read-ahead: same object, same index, failures in order", [&] {
        for (int mode = 1; mode < 4; mode++) {
            const bool comp = (mode & 1) != 0, enc = (mode & 2) != 0;
            auto make = [&](bool ahead) {
                return std::make_shared<GpuTransformChunkEnumeration>(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), 4096 + 3), comp,
                    enc ? std::optional<DataKeyAndAAD>(DataKeyAndAAD{KEY, AAD}) : std::nullopt, countingIv(), 5, true, TSX_ZSTD_PROFILE_1_5_7, ahead);
            };
            auto a = make(false), b = make(true);
            TransformFinisher fa(a, (int)text.size()), fb(b, (int)text.size());
            CHECK(fa.toBytes() == fb.toBytes());
            CHECK(a->crc32cOfOriginalChunks() == b->crc32cOfOriginalChunks() && !a->crc32cOfOriginalChunks().empty());
            auto ia = fa.chunkIndex(), ib = fb.chunkIndex();
            CHECK(ia->chunks().size() == ib->chunks().size());
            for (size_t i = 0; i < ia->chunks().size(); i++) CHECK(ia->chunks()[i].transformedSize == ib->chunks()[i].transformedSize);
        }
        auto n = std::make_shared<int>(0);
        IvSupplier failing = [n](uint8_t iv[12]) { if ((*n)++ == 4) throw std::runtime_error("entropy source failed"); memset(iv, 7, 12); };
        GpuTransformChunkEnumeration e(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), 4096), false, DataKeyAndAAD{KEY, AAD}, failing, 3, false,
                                       TSX_ZSTD_PROFILE_1_5_7, true);
        for (int i = 0; i < 3; i++) { CHECK(e.hasMoreElements()); CHECK(e.nextElement().size() == 4096 + 28); }     // batch 0 (IVs 0..2) is whole
        expectThrows<std::runtime_error>([&] { e.hasMoreElements(); }, "entropy source failed");                   // batch 1 (IV 4) is not
        {
            GpuTransformChunkEnumeration dropped(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), 4096), true, DataKeyAndAAD{KEY, AAD}, countingIv(), 4, false,
                                                 TSX_ZSTD_PROFILE_1_5_7, true);
            CHECK(dropped.nextElement().size() > 28);                                                               // the helper is working on batch 1 now
        }
    });
    run("full chain bytes == oracle chain (Zstd frame + GCM) for the supplier's IVs", [&] {
        if (std::string(orc_zstd_version()).rfind("1.5.7", 0) != 0) { printf("    (skipped: libzstd 1.5.7 not available)\n"); return; }
        const int cs = full ? 1 << 20 : 20000;
        GpuTransformChunkEnumeration e(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), cs), true, DataKeyAndAAD{KEY, AAD}, countingIv(), 3, true);
        size_t pos = 0; uint64_t n = 0;
        while (e.hasMoreElements()) {
            const Bytes got = e.nextElement();
            const size_t len = std::min((size_t)cs, text.size() - pos);
            Bytes exp(orc_chain_bound(len, 7)); uint32_t crc; uint8_t iv[12] = {0}; for (int i = 0; i < 8; i++) iv[11 - i] = (uint8_t)(n >> (8 * i));
            Bytes scratch(orc_chain_bound(len, 1) + 64);
            exp.resize(orc_transform_chunk(7 /*compress|encrypt|crc*/, KEY.data(), AAD.data(), AAD.size(), iv, text.data() + pos, len, exp.data(), exp.size(), scratch.data(), &crc));
            CHECK(got == exp);
            CHECK(e.crc32cOfOriginalChunks()[n] == crc);
            pos += len; n++;
        }
        CHECK(pos == text.size());
    });
    run("DecryptionChunkEnumerationTest: wrong AAD / flipped byte -> \"Tag mismatch\"; earlier chunks still delivered", [&] {
        GpuTransformChunkEnumeration enc(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), 8192), false, DataKeyAndAAD{KEY, AAD}, countingIv(), 8);
        TransformFinisher tf(std::shared_ptr<TransformChunkEnumeration>(&enc, [](TransformChunkEnumeration*) {}), (int)text.size());
        Bytes up = tf.toBytes(); auto index = tf.chunkIndex();
        up[(size_t)index->chunks()[2].transformedPosition + 100] ^= 1;                              // corrupt chunk 2
        GpuDetransformChunkEnumeration d(be, std::make_shared<BaseDetransformChunkEnumeration>(stream(up), index->chunks()), false, SegmentEncryptionMetadata{KEY, AAD, 12}, 8192, 8);
        CHECK(d.nextElement() == Bytes(text.begin(), text.begin() + 8192));
        CHECK(d.nextElement() == Bytes(text.begin() + 8192, text.begin() + 16384));
        expectThrows<std::runtime_error>([&] { d.nextElement(); }, "Tag mismatch");
        Bytes badAad = AAD; badAad[0] ^= 1;
        GpuDetransformChunkEnumeration d2(be, std::make_shared<BaseDetransformChunkEnumeration>(stream(tf.chunkIndex() ? up : up), index->chunks()), false, SegmentEncryptionMetadata{KEY, badAad, 12}, 8192, 8);
        expectThrows<std::runtime_error>([&] { d2.nextElement(); }, "Tag mismatch");
    });
    run("DecompressionChunkEnumeration: frame without content size -> \"Invalid decompressed size: -1\"", [&] {
        const Bytes noSize = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58, 0x01, 0x00, 0x00};
        GpuDetransformChunkEnumeration d(be, std::make_shared<BaseDetransformChunkEnumeration>(stream(noSize), std::vector<Chunk>{Chunk{0, 0, 16, 0, 9}}), true, std::nullopt, 16, 1);
        expectThrows<std::runtime_error>([&] { d.nextElement(); }, "Invalid decompressed size: -1");
    });
    // CT/manifest/index/ChunkIndexSerializationTest.java:39-123 — the reference's only golden Zstd frame
    run("ChunkIndexSerializationTest: ENCODED_CHUNKS + JSON of both index kinds", [&] {
        CHECK(serializeTransformedChunks(*be, {10, 20, 30}) == "KLUv/SAPeQAAAAAAAwAAAAoBAAoAAAAe");
        CHECK((deserializeTransformedChunks(*be, "KLUv/SAPeQAAAAAAAwAAAAoBAAoAAAAe") == std::vector<int>{10, 20, 30}));
        FixedSizeChunkIndex fixed(100, 250, 110, 30);
        CHECK(chunkIndexToJson(*be, fixed) == "{\"type\":\"fixed\",\"originalChunkSize\":100,\"originalFileSize\":250,\"transformedChunkSize\":110,\"finalTransformedChunkSize\":30}");
        VariableSizeChunkIndex var(100, 250, {10, 20, 30});
        const std::string vj = "{\"type\":\"variable\",\"originalChunkSize\":100,\"originalFileSize\":250,\"transformedChunks\":\"KLUv/SAPeQAAAAAAAwAAAAoBAAoAAAAe\"}";
        CHECK(chunkIndexToJson(*be, var) == vj);
        CHECK((chunkIndexFromJson(*be, vj)->chunks() == std::vector<Chunk>{{0, 0, 100, 0, 10}, {1, 100, 100, 10, 20}, {2, 200, 50, 30, 30}}));
        CHECK((chunkIndexFromJson(*be, chunkIndexToJson(*be, fixed))->chunks() == std::vector<Chunk>{{0, 0, 100, 0, 110}, {1, 100, 100, 110, 110}, {2, 200, 50, 220, 30}}));
        std::vector<int> many; std::mt19937 r(5); for (int i = 0; i < 2000; i++) many.push_back(1000000 + (int)(r() % 300));      // README.md:162-170 sized index
        CHECK(deserializeTransformedChunks(*be, serializeTransformedChunks(*be, many)) == many);
    });
    // upload sink: TransformFinisher.toInputStream() -> ObjectUploader.upload (RemoteStorageManager.java:410-420,
    // FileSystemStorage.java:51-60), then every chunk back through GpuChunkManager from the stored object
    run("upload .log object to FileSystemStorage through the chain, fetch chunks back", [&] {
        char tmpl[] = "/tmp/tsxhost_fs_XXXXXX";
        CHECK(mkdtemp(tmpl) != nullptr);
        auto fs = std::make_shared<FileSystemStorage>(tmpl);
        expectThrows<std::invalid_argument>([] { FileSystemStorage bad("/nonexistent/dir"); }, "/nonexistent/dir must be a writable directory");
        const int cs = 32768;
        auto t = std::make_shared<GpuTransformChunkEnumeration>(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), cs), true, DataKeyAndAAD{KEY, AAD}, countingIv(), 8);
        TransformFinisher tf(t, (int)text.size());
        const long bytes = fs->upload(*tf.toInputStream(), "topic-abc/7/00000000000000000023-segment.log");
        SegmentManifest m; m.chunkIndex = tf.chunkIndex(); m.compression = true; m.encryption = SegmentEncryptionMetadata{KEY, AAD, 12};
        const auto& chunks = m.chunkIndex->chunks();
        CHECK(bytes == chunks.back().transformedPosition + chunks.back().transformedSize);
        GpuChunkManager cm(be, fs);
        Bytes all;
        for (size_t id = 0; id < chunks.size(); id += 3) {
            const auto part = cm.getChunks("topic-abc/7/00000000000000000023-segment.log", m, (int)id, (int)std::min<size_t>(3, chunks.size() - id));
            for (const auto& c : part) all.insert(all.end(), c.begin(), c.end());
        }
        CHECK(all == text);
    });
    run("RateLimitedInputStream: bucket of `rate` tokens, refilled greedily (RateLimitedInputStream.java:46-84)", [&] {
        const Bytes data = randomBytes(40 * 1024, 3);
        auto bucket = std::make_shared<TokenBucket>(16384);
        RateLimitedInputStream in(stream(data), bucket);
        timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
        const Bytes got = in.readAllBytes();                              // 40 KiB at 16 KiB/s with a full 16 KiB bucket: >= 1.4 s
        clock_gettime(CLOCK_MONOTONIC, &b);
        const double el = (double)(b.tv_sec - a.tv_sec) + (double)(b.tv_nsec - a.tv_nsec) * 1e-9;
        CHECK(got == data);
        CHECK(el > 1.2 && el < 3.0);
        TokenBucket minRate(1);                                           // below MIN_RATE the bucket is built with MIN_RATE
        timespec c; clock_gettime(CLOCK_MONOTONIC, &a); minRate.consume(8192); clock_gettime(CLOCK_MONOTONIC, &c);
        CHECK((double)(c.tv_sec - a.tv_sec) + (double)(c.tv_nsec - a.tv_nsec) * 1e-9 < 0.2);
    });
    // C/SegmentCompressionChecker.java:37-53 (+ Kafka DefaultRecordBatch.ensureValid) and C/RemoteStorageManager.java:455-490
    run("SegmentCompressionChecker.check + transformIndex", [&] {
        auto batch = [&](uint16_t attributes, size_t payload) {
            Bytes b(61 + payload, 0x5A);
            const uint32_t batchLength = (uint32_t)(b.size() - 12);
            for (int i = 0; i < 8; i++) b[(size_t)i] = 0;                              // baseOffset
            b[8] = (uint8_t)(batchLength >> 24); b[9] = (uint8_t)(batchLength >> 16); b[10] = (uint8_t)(batchLength >> 8); b[11] = (uint8_t)batchLength;
            b[16] = 2; b[21] = (uint8_t)(attributes >> 8); b[22] = (uint8_t)attributes;
            const uint32_t crc = orc_crc32c(b.data() + 21, b.size() - 21);
            b[17] = (uint8_t)(crc >> 24); b[18] = (uint8_t)(crc >> 16); b[19] = (uint8_t)(crc >> 8); b[20] = (uint8_t)crc;
            return b;
        };
        Bytes plain = batch(0, 5000), zstdBatch = batch(4, 333);
        plain.insert(plain.end(), 100, 0x11);                                          // more batches may follow in the segment
        CHECK(!segmentIsCompressed(*be, plain) && segmentIsCompressed(*be, zstdBatch));
        Bytes corrupt = plain; corrupt[100] ^= 1;
        try { segmentIsCompressed(*be, corrupt); CHECK(false); } catch (const InvalidRecordBatchException& e) { CHECK(std::string(e.what()).rfind("Record is corrupt (stored crc = ", 0) == 0); }
        try { segmentIsCompressed(*be, Bytes(plain.begin(), plain.begin() + 40)); CHECK(false); } catch (const InvalidRecordBatchException&) {}
        // index files: one chunk, encrypted only (CIT RemoteStorageManagerTest.java:472-596: size recorded = transformed size)
        const Bytes index = randomBytes(10 * 1024 + 7, 9);
        CHECK(transformIndex(be, index, std::nullopt) == index && transformIndex(be, {}, DataKeyAndAAD{KEY, AAD}).empty());
        const Bytes enc = transformIndex(be, index, DataKeyAndAAD{KEY, AAD}, countingIv());
        CHECK(enc.size() == index.size() + 28);
        Bytes dec(index.size());
        CHECK(orc_gcm_decrypt_chunk(KEY.data(), AAD.data(), AAD.size(), enc.data(), enc.size(), dec.data()) == (long)index.size() && dec == index);
    });
    // C/fetch/DefaultChunkManager.java:50-70 — every chunk by id through GpuChunkManager, and a prefetch window in one batch
    run("ChunkManager.getChunk / getChunks over an uploaded object", [&] {
        const int cs = 16384;
        GpuTransformChunkEnumeration t(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), cs), true, DataKeyAndAAD{KEY, AAD}, countingIv(), 32);
        TransformFinisher tf(std::shared_ptr<TransformChunkEnumeration>(&t, [](TransformChunkEnumeration*) {}), (int)text.size());
        auto fetcher = std::make_shared<MemFetcher>(); fetcher->object = tf.toBytes();
        SegmentManifest m; m.chunkIndex = tf.chunkIndex(); m.compression = true; m.encryption = SegmentEncryptionMetadata{KEY, AAD, 12};
        GpuChunkManager cm(be, fetcher);
        const auto& chunks = m.chunkIndex->chunks();
        for (int id : {0, 1, (int)chunks.size() / 2, (int)chunks.size() - 1}) {
            const Chunk& c = chunks[(size_t)id];
            CHECK(cm.getChunk("k.log", m, id) == Bytes(text.begin() + c.originalPosition, text.begin() + c.originalPosition + c.originalSize));
        }
        fetcher->fetches = 0;
        const auto win = cm.getChunks("k.log", m, 1, 4);                                              // 4-chunk prefetch window: one ranged fetch, one batch
        CHECK(fetcher->fetches == 1 && win.size() == 4);
        for (int k = 0; k < 4; k++) { const Chunk& c = chunks[(size_t)(1 + k)]; CHECK(win[(size_t)k] == Bytes(text.begin() + c.originalPosition, text.begin() + c.originalPosition + c.originalSize)); }
    });
    // CT/manifest/SegmentManifestV1SerdeTest.java:40-133 - the three golden strings
    run("SegmentManifestV1SerdeTest: withEncryption / withoutEncryption / withoutTxnIndex", [&] {
        const std::string rlsm = "{\"remoteLogSegmentId\":{\"topicIdPartition\":{\"topicId\":\"lZ6vvmajTWKDBUTV6SQAtQ\",\"topicPartition\":"
                                 "{\"topic\":\"topic1\",\"partition\":42}},\"id\":\"adh9f8BMS4anaUnD8KWfWg\"},\"startOffset\":0,\"endOffset\":1000,"
                                 "\"maxTimestampMs\":1000000000,\"brokerId\":2,\"eventTimestampMs\":2000000000,"
                                 "\"segmentLeaderEpochs\":{\"0\":100,\"1\":200,\"2\":300}}";
        const std::string head = "{\"version\":\"1\",\"chunkIndex\":{\"type\":\"fixed\",\"originalChunkSize\":100,\"originalFileSize\":1000,"
                                 "\"transformedChunkSize\":110,\"finalTransformedChunkSize\":110},\"segmentIndexes\":{\"offset\":{\"position\":0,\"size\":1},"
                                 "\"timestamp\":{\"position\":1,\"size\":1},\"producerSnapshot\":{\"position\":2,\"size\":1},\"leaderEpoch\":{\"position\":3,\"size\":1},";
        const std::string withoutEnc = head + "\"transaction\":{\"position\":4,\"size\":1}},\"compression\":false,\"remoteLogSegmentMetadata\":" + rlsm + "}";
        const std::string withoutTxn = head + "\"transaction\":null},\"compression\":false,\"remoteLogSegmentMetadata\":" + rlsm + "}";
        // with encryption the reference compares after removing the (random, RSA-wrapped) dataKey: what remains is {"aad":"CgsMDQ=="}
        const std::string withEncNoKey = head + "\"transaction\":{\"position\":4,\"size\":1}},\"compression\":false,\"encryption\":{\"aad\":\"CgsMDQ==\"},"
                                         "\"remoteLogSegmentMetadata\":" + rlsm + "}";
        SegmentManifestV1 m;
        m.chunkIndex = std::make_shared<FixedSizeChunkIndex>(100, 1000, 110, 110);
        m.segmentIndexes = SegmentIndexesV1{{0, 1}, {1, 1}, {2, 1}, {3, 1}, SegmentIndexV1{4, 1}};
        m.compression = false; m.remoteLogSegmentMetadataJson = rlsm;
        CHECK(segmentManifestToJson(*be, m) == withoutEnc);
        SegmentManifestV1 back = segmentManifestFromJson(*be, withoutEnc);
        CHECK(back.chunkIndex->chunks() == m.chunkIndex->chunks() && !back.compression && !back.encryption && back.segmentIndexes.transaction.has_value());
        CHECK(back.segmentIndexes.offset == (SegmentIndexV1{0, 1}) && back.segmentIndexes.leaderEpoch == (SegmentIndexV1{3, 1}) && *back.segmentIndexes.transaction == (SegmentIndexV1{4, 1}));
        CHECK(segmentManifestToJson(*be, back) == withoutEnc);                                        // parse -> write is the identity on the golden
        SegmentManifestV1 noTxn = m; noTxn.segmentIndexes.transaction.reset();
        CHECK(segmentManifestToJson(*be, noTxn) == withoutTxn);
        CHECK(!segmentManifestFromJson(*be, withoutTxn).segmentIndexes.transaction.has_value());
        SegmentManifestV1 enc = m;
        enc.encryption = SegmentEncryptionMetadata{Bytes{0, 1, 2, 3, 4, 5, 6, 7, 8, 9}, Bytes{10, 11, 12, 13}, 12};       // DATA_KEY, AAD of the reference test
        const std::string wrapped = "static-key-id:" + base64Encode(Bytes{9, 8, 7, 6});               // stands for RsaEncryptionProvider.encryptDataKey
        const std::string ej = segmentManifestToJson(*be, enc, [&](const Bytes& k) { CHECK(k.size() == 10); return wrapped; });
        const std::string keyProp = "\"dataKey\":\"" + wrapped + "\",";
        const size_t at = ej.find(keyProp);
        CHECK(at != std::string::npos);
        std::string without = ej; without.erase(at, keyProp.size());
        CHECK(without == withEncNoKey);
        SegmentManifestV1 eb = segmentManifestFromJson(*be, ej, [&](const std::string& sk) { CHECK(sk == wrapped); return Bytes{0, 1, 2, 3, 4, 5, 6, 7, 8, 9}; });
        CHECK(eb.encryption && eb.encryption->aad == (Bytes{10, 11, 12, 13}) && eb.encryption->dataKey.size() == 10);
        expectThrows<std::invalid_argument>([&] { segmentManifestToJson(*be, enc); }, "a data-key encryptor is required to serialise an encrypted segment's manifest");
        expectThrows<std::invalid_argument>([&] { segmentManifestFromJson(*be, "{\"version\":\"2\"}"); }, "Could not resolve type id '2' as a subtype of SegmentManifest");
        expectThrows<std::invalid_argument>([&] { segmentManifestFromJson(*be, "{\"version\":\"1\",\"chunkIndex\":{\"type\":\"fixed\",\"originalChunkSize\":1,\"originalFileSize\":1,\"transformedChunkSize\":1,\"finalTransformedChunkSize\":1}}"); },
                                            "Missing required creator property 'segmentIndexes'");
        // a variable index (compressed segment) inside a manifest: the index's Zstd frame goes through the device compressor
        SegmentManifestV1 var = m; var.compression = true; var.chunkIndex = std::make_shared<VariableSizeChunkIndex>(100, 250, std::vector<int>{10, 20, 30});
        const std::string vj = segmentManifestToJson(*be, var);
        CHECK(vj.find("\"transformedChunks\":\"KLUv/SAPeQAAAAAAAwAAAAoBAAoAAAAe\"") != std::string::npos && vj.find("\"compression\":true") != std::string::npos);
        CHECK(segmentManifestFromJson(*be, vj).chunkIndex->chunks() == var.chunkIndex->chunks());
    });
    // C/fetch/cache/ChunkCache.java:76-129,159-184 with the device in mind: a window is ONE fetch + ONE device batch
    run("GpuChunkCache: prefetch window and concurrent misses coalesce into one fetch + one batch", [&] {
        const int cs = 4096;
        GpuTransformChunkEnumeration t(be, std::make_shared<BaseTransformChunkEnumeration>(stream(text), cs), true, DataKeyAndAAD{KEY, AAD}, countingIv(), 32);
        TransformFinisher tf(std::shared_ptr<TransformChunkEnumeration>(&t, [](TransformChunkEnumeration*) {}), (int)text.size());
        auto fetcher = std::make_shared<MemFetcher>(); fetcher->object = tf.toBytes();
        SegmentManifest m; m.chunkIndex = tf.chunkIndex(); m.compression = true; m.encryption = SegmentEncryptionMetadata{KEY, AAD, 12};
        const auto& chunks = m.chunkIndex->chunks();
        auto plain = [&](int id) { const Chunk& c = chunks[(size_t)id]; return Bytes(text.begin() + c.originalPosition, text.begin() + c.originalPosition + c.originalSize); };
        CHECK(chunks.size() >= 12);
        {   // one caller, prefetch of 3 chunks: chunk 2 and chunks 3..5 leave as ONE getChunks; 3, 4, 5 are hits afterwards
            GpuChunkCache cache(std::make_shared<GpuChunkManager>(be, fetcher), 3 * cs, (size_t)64 << 20, 10000, 0);
            fetcher->fetches = 0;
            CHECK(cache.getChunk("k.log", m, 2) == plain(2));
            CHECK(fetcher->fetches == 1 && cache.stats().fetchCalls == 1 && cache.stats().chunksFetched == 4);
            CHECK(cache.getChunk("k.log", m, 3) == plain(3) && cache.getChunk("k.log", m, 4) == plain(4));
            // chunk 3 prefetched 6, chunk 4 prefetched 7: one chunk each, nothing beyond the window of the chunk asked for
            cache.quiesce();
            ChunkCacheStats s1 = cache.stats();
            CHECK(s1.misses == 1 && s1.hits == 2 && s1.chunksFetched == 6);
            CHECK(cache.getChunk("k.log", m, (int)chunks.size() - 1) == plain((int)chunks.size() - 1));      // last chunk: no window behind it
        }
        {   // 8 threads miss chunks 0..7 of the same object at once: they meet in one batch (a few at most), not in 8
            // (the leader waits 1 s here: under ThreadSanitizer, or next to a compiler that keeps all cores busy, eight threads have needed more than 250 ms)
            GpuChunkCache cache(std::make_shared<GpuChunkManager>(be, fetcher), 0, (size_t)64 << 20, 10000, 1000000);
            fetcher->fetches = 0;
            std::vector<std::thread> th; std::vector<int> ok(8, 0);
            std::atomic<int> go{0};
            for (int i = 0; i < 8; i++) th.emplace_back([&, i] {
                go++; while (go.load() < 8) std::this_thread::yield();
                while (cache.stats().misses < i) std::this_thread::yield();             // arrive IN ORDER (a miss joins the open batch only when it continues it): thread i goes in when the
                                                                                          // misses of 0 .. i - 1 are registered - sleeps of 5 ms x i swapped two threads next to a busy compiler
                ok[(size_t)i] = cache.getChunk("k.log", m, i) == plain(i);
            });
            for (auto& x : th) x.join();
            CHECK(std::all_of(ok.begin(), ok.end(), [](int v) { return v == 1; }));
            const ChunkCacheStats s2 = cache.stats();
            CHECK(s2.misses == 8 && s2.chunksFetched == 8 && s2.fetchCalls <= 3 && fetcher->fetches == s2.fetchCalls && s2.joined >= 5);
        }
        {   // a forged chunk fails only its own callers: the window is retried chunk by chunk
            auto bad = std::make_shared<MemFetcher>(); bad->object = fetcher->object;
            bad->object[(size_t)chunks[5].transformedPosition + 20] ^= 1;
            GpuChunkCache cache(std::make_shared<GpuChunkManager>(be, bad), 3 * cs, (size_t)64 << 20, 10000, 0);
            CHECK(cache.getChunk("k.log", m, 4) == plain(4));                                        // window 4..7 contains the forged chunk 5
            expectThrows<std::runtime_error>([&] { cache.getChunk("k.log", m, 5); }, "Tag mismatch");
            CHECK(cache.getChunk("k.log", m, 6) == plain(6));
        }
        {   // a load that dies of something that is not a std::exception (the Java twin's OutOfMemoryError / UnsatisfiedLinkError):
            // every waiter of the window is failed at once - nobody sits out get.timeout.ms - and the ids are loadable again afterwards
            struct Dying : ObjectFetcher {
                std::shared_ptr<ObjectFetcher> real; std::atomic<bool> die{true};
                std::shared_ptr<InputStream> fetch(const std::string& k, BytesRange r) override { if (die) throw 42; return real->fetch(k, r); }
            };
            auto dying = std::make_shared<Dying>(); dying->real = fetcher;
            GpuChunkCache cache(std::make_shared<GpuChunkManager>(be, dying), 3 * cs, (size_t)64 << 20, 10000, 0);
            const auto t0 = std::chrono::steady_clock::now();
            bool threw = false;
            try { cache.getChunk("k.log", m, 2); } catch (...) { threw = true; }
            CHECK(threw && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5));
            cache.quiesce();
            dying->die = false;
            CHECK(cache.getChunk("k.log", m, 2) == plain(2) && cache.getChunk("k.log", m, 3) == plain(3));       // nothing of the dead load is left in `pending`
        }
        {   // fetchLogSegment's caller on top: an original-offset range through FetchChunkEnumeration -> GpuChunkCache -> GpuChunkManager
            // gives exactly those bytes, and a reader that closes early has caused nothing beyond its chunk's window to be fetched
            auto cache = std::make_shared<GpuChunkCache>(std::make_shared<GpuChunkManager>(be, fetcher), 2 * cs, (size_t)64 << 20, 10000, 0);
            const int from = cs + 100, to = 5 * cs + 17;
            FetchChunkEnumeration e(cache, "k.log", m, BytesRange{from, to});
            CHECK(e.startChunkId() == 1 && e.lastChunkId() == 5);
            CHECK(e.readAll() == Bytes(text.begin() + from, text.begin() + to + 1));
            cache->quiesce();
            CHECK(cache->stats().chunksFetched <= 8);                                    // chunks 1..5 + at most the 2-chunk window behind chunk 5
            auto cache2 = std::make_shared<GpuChunkCache>(std::make_shared<GpuChunkManager>(be, fetcher), 2 * cs, (size_t)64 << 20, 10000, 0);
            FetchChunkEnumeration early(cache2, "k.log", m, BytesRange{0, (int)text.size() - 1});
            CHECK(early.nextElement() == plain(0));
            early.close();
            cache2->quiesce();
            CHECK(!early.hasMoreElements() && cache2->stats().chunksFetched == 3);       // chunk 0 + its window (1, 2): nothing further
        }
        {   // eviction by weight: a cache of two chunks keeps two
            GpuChunkCache cache(std::make_shared<GpuChunkManager>(be, fetcher), 0, (size_t)2 * cs, 10000, 0);
            for (int i = 0; i < 5; i++) CHECK(cache.getChunk("k.log", m, i) == plain(i));
            CHECK(cache.stats().evictions == 3);
            CHECK(cache.getChunk("k.log", m, 4) == plain(4) && cache.stats().hits == 1);
        }
    });
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 2) { printf("usage: host_tests cpu | backend <libtsxform path> [full] | segment <libtsxform path> <file>\n"); return 2; }
    if (std::string(argv[1]) == "segment") {
        // SegmentCompressionChecker.check on the head of a segment file someone else made (tests/test_synth_b.py: content "B" of synth.py):
        // prints what the twin says - "compressed=0|1" - or the exception's text
        if (argc < 4) return 2;
        try {
            Backend be(argv[2]);
            FILE* f = fopen(argv[3], "rb"); if (!f) { printf("cannot open %s\n", argv[3]); return 2; }
            Bytes head((size_t)1 << 20); head.resize(fread(head.data(), 1, head.size(), f)); fclose(f);
            printf("compressed=%d\n", segmentIsCompressed(be, head) ? 1 : 0);
            return 0;
        } catch (const InvalidRecordBatchException& e) { printf("InvalidRecordBatchException: %s\n", e.what()); return 3; }
          catch (const std::exception& e) { printf("error: %s\n", e.what()); return 4; }
    }
    if (std::string(argv[1]) == "cpu") { cpuTests(); segmentIndexesBuilderTests(); fetchEnumerationTests(); }
    else { if (argc < 3) return 2; g_lib = argv[2]; try { backendTests(argc > 3 && std::string(argv[3]) == "full"); } catch (const std::exception& e) { printf("  FAIL backend: %s\n", e.what()); g_failed++; } }
    printf("%d run, %d failed\n", g_run, g_failed);
    return g_failed ? 1 : 0;
}
