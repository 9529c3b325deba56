"""sha256 over the kernel sources (rendering_amd/csrc/*): stamps the PMC / ISA summaries under profiles/ so that bench.py
can tell whether they belong to the kernels it is running."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rendering_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")):
            h.update(n.encode())
            h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
