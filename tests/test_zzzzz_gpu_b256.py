"""Content "B" (Kafka v2 record batches, tsxform/synth.py) at the size VERDICT r4 #7 asks of the device parity set: 256 chunks in one batch.
(A file of its own, last in the alphabet: it was written after the round's last GPU call - the batch itself passed on the CPU emulator of the same
kernel sources, profiles/r05_b256_on_the_emulator.txt.)"""
import pytest

from tests import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_256_chunks_of_binary_content_both_profiles(gpu, oracle):
    """256 chunks of content "B" in one batch: full chain = libzstd 1.5.7 + OpenSSL byte for byte, round trip, and both Zstd profiles (the
    pre-splitter of 1.5.7 cuts on many of them: there the profiles' frames differ and both decode; elsewhere profile 1.5.6 is pinned to the real
    library).  The same batch ran through the CPU emulator of the kernel sources (profiles/r05_b256_on_the_emulator.txt)."""
    if not oracle.zstd_version().startswith("1.5.7"):
        pytest.skip("libzstd 1.5.7 not available")
    ratio, differ, pinned = pc.check_b_batch_256(gpu, oracle)
    print("B x 256: transformed / original = %.3f; the pre-splitter cut on %d chunks, %d pinned to the real library under profile 1.5.6" % (ratio, differ, pinned))
    assert differ >= 16 and differ + pinned == 256
