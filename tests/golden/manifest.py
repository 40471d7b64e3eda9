#!/usr/bin/env python
"""Hash manifest of the committed golden vectors and which generator wrote each.

    python tests/golden/manifest.py            # verify tests/golden/MANIFEST.sha256 against the files
    python tests/golden/manifest.py --write    # rewrite it (after regenerating a golden with its script)

The goldens stay IN the repository (89 MB): they are outputs of the reference's own Python / CUDA sources, which exist only in the
build container (/root/reference) -- the GPU box that runs `pytest -m gpu` cannot regenerate them.  The manifest pins what the tests
compare against: a regenerated file that differs shows up here, not as a silently moved tolerance."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GENERATOR = {
    "ba_python.npz": "make_golden.py", "corr_python.npz": "make_golden.py", "update_python.npz": "make_golden.py",
    "update_autocast_python.npz": "make_golden.py", "encoder_python.npz": "make_golden.py",
    "graph_python.npz": "make_graph_golden.py",
    "graph_c2_python.npz": "make_graph_scale_golden.py c2", "graph_c3_python.npz": "make_graph_scale_golden.py c3",
    "graph_stereo_python.npz": "make_graph_scale_golden.py stereo", "graph_tum_size_python.npz": "make_graph_scale_golden.py tum",
    "graph_wide_python.npz": "make_graph_scale_golden.py wide", "graph_big_python.npz": "make_graph_scale_golden.py big",
    "graph_c5_python.npz": "make_graph_scale_golden.py c5 (reference update_lowmem at 1024 keyframes / 8192 edges, stereo + sensor depth + confidence map; ~30 min)",
    "graph_scale_probe.json": "make_graph_scale_golden.py c2 c3 c5 stereo tum wide big --probe",
    "policy_python.npz": "make_policy_golden.py", "ref_cuda.npz": "make_ref_golden.py (on an MI355X: the reference's CUDA sources compiled for gfx950)",
}


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def current():
    return {n: sha(os.path.join(HERE, n)) for n in sorted(GENERATOR) if os.path.exists(os.path.join(HERE, n))}


def read():
    out = {}
    for line in open(os.path.join(HERE, "MANIFEST.sha256")):
        if line.strip() and not line.startswith("#"):
            h, n = line.split()[:2]
            out[n] = h
    return out


if __name__ == "__main__":
    if "--write" in sys.argv:
        with open(os.path.join(HERE, "MANIFEST.sha256"), "w") as f:
            f.write("# sha256  file  <- generator (tests/golden/manifest.py --write)\n")
            for n, h in current().items():
                f.write("%s  %s  <- %s\n" % (h, n, GENERATOR[n]))
        print("written")
    else:
        cur, want = current(), read()
        bad = [n for n in sorted(set(cur) | set(want)) if cur.get(n) != want.get(n)]
        print("ok" if not bad else "MISMATCH: %s" % bad)
        sys.exit(1 if bad else 0)
