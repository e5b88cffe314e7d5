#!/bin/bash
# build an experiment flavour of the product library: tools/build_variant.sh <tag> <extra -D flags...>  ->  tools/_libs/libtsxform_<tag>.so
set -e
tag=$1; shift
cd /root/repo/tiered-storage-for-apache-kafka_amd/csrc
mkdir -p _obj/var_$tag
for f in tsx_api crc32c gcm zstd_enc zstd_dec; do
  if [ $f = zstd_enc ] || [ ! -f _obj/var_base_$f.o ]; then
    out=_obj/var_${tag}/$f.o
    [ $f != zstd_enc ] && out=_obj/var_base_$f.o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value "$@" -c $f.hip -o $out
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_libs/libtsxform_$tag.so _obj/var_base_tsx_api.o _obj/var_base_crc32c.o _obj/var_base_gcm.o _obj/var_${tag}/zstd_enc.o _obj/var_base_zstd_dec.o
rm -f ../../tools/_libs/libtsxform_$tag.so.*
