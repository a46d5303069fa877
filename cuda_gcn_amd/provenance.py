"""Which tree a committed measurement belongs to: sha256 of the kernel sources it depends on, stored in the file's `_meta`
by the tools that write profiles/*.json and compared by bench.py when it quotes such a file beside a live timing
(a file whose sources differ from the tree's is quoted with `stale: true`)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda_gcn_amd", "csrc")


def source_sha(names):
    """{name: first 16 hex digits of the sha256 of cuda_gcn_amd/csrc/<name> (or of a path relative to the repo root)}"""
    out = {}
    for n in names:
        p = os.path.join(CSRC, n) if os.path.exists(os.path.join(CSRC, n)) else os.path.join(ROOT, n)
        try:
            out[n] = hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
        except OSError:
            out[n] = None
    return out


def stale_reason(meta):
    """None when every source recorded in `meta["sources"]` still has the recorded hash; else why the file is stale"""
    rec = (meta or {}).get("sources")
    if not rec:
        return "the file records no source hashes (taken before round 6)"
    now = source_sha(list(rec))
    diff = [n for n in rec if rec[n] != now.get(n)]
    return None if not diff else "changed since the file was taken: " + ", ".join(diff)
