"""sha256 over the kernel / engine sources: profiles record it, bench.py compares it with the tree it runs from (a PMC table measured on
another build must not be quoted as this build's traffic)."""
import glob, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest() -> str:
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "wacv23_tsnet_amd", "csrc", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(digest())
