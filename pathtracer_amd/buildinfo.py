"""Identity of the device code a measurement belongs to: git blob hashes of the kernel sources.

``scripts/pmc_digest.py`` stores them beside the counter digests in ``profiles/pmc.json``; ``bench.py`` compares
them with the working tree and marks the counter-derived roofline figures stale when a kernel source has
changed since the counters were collected."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
KERNEL_SOURCES = ("pt_device.h", "pt_math.hip.h", "pt_traverse.hip.h", "pt_shade.hip.h", "pt_kernels.hip")


def git_blob_hash(path: str) -> str:
    """What `git hash-object` prints for the file."""
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_source_hashes() -> dict:
    return {name: git_blob_hash(os.path.join(CSRC, name)) for name in KERNEL_SOURCES}
