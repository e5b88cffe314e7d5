#!/bin/bash
# build the emulator flavour + product libs, run the emulator zstd tests
cd /root/repo/tiered-storage-for-apache-kafka_amd/csrc && make emu 2>&1 | grep -E " error|error:"
cd /root/repo && timeout 1500 python -m pytest tests/test_emu_zstd.py -x -q 2>&1 | tail -4
