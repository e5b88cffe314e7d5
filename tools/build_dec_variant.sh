#!/bin/bash
# decoder experiment flavours: tools/build_dec_variant.sh <tag> <patch files...> -> tiered-storage-for-apache-kafka_amd/libtsxform_dec_<tag>.so
# (zstd_dec.hip with the patches applied in a scratch copy; every other object from the regular build)
set -e
tag=$1; shift
ROOT=/root/repo; C=$ROOT/tiered-storage-for-apache-kafka_amd/csrc
make -s -C $C > /dev/null
tmp=$(mktemp -d); mkdir -p $tmp/tiered-storage-for-apache-kafka_amd/csrc
cp $C/*.h $C/zstd_dec.hip $tmp/tiered-storage-for-apache-kafka_amd/csrc/; mkdir -p $tmp/include; cp $ROOT/include/tsxform.h $tmp/include/
for p in "$@"; do (cd $tmp && patch -s -p1 < $p); done
(cd $tmp/tiered-storage-for-apache-kafka_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -c zstd_dec.hip -o $C/_obj/dec_$tag.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tiered-storage-for-apache-kafka_amd/libtsxform_dec_$tag.so $C/_obj/tsx_api.o $C/_obj/crc32c.o $C/_obj/gcm.o $C/_obj/zstd_enc.o $C/_obj/dec_$tag.o
rm -rf $tmp; ls -la $ROOT/tiered-storage-for-apache-kafka_amd/libtsxform_dec_$tag.so
