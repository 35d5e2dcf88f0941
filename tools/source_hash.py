#!/usr/bin/env python3
"""CPU.  sha256 over the sources librt_amd is built from (rust-tracing_amd/csrc/*, include/rt_amd*.h): what ties a profile
to a build.  tools/profile_round.sh records it on the GPU box (which has no .git), tools/collect_profiles.py refuses to file a
profile whose hash is not the working tree's, bench.py quotes committed counters only for the sources it runs."""
import hashlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def source_hash() -> str:
    h = hashlib.sha256()
    files = sorted(p for d, pattern in (("rust-tracing_amd/csrc", "*"), ("include", "rt_amd*.h")) for p in (ROOT / d).glob(pattern)
                   if p.is_file() and p.suffix in (".hip", ".h", ".hpp", ".cpp"))
    for p in files:
        h.update(p.name.encode()); h.update(b"\0"); h.update(p.read_bytes()); h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
