"""Loader timing, CPU only, run in the build container where oracle/_ref exists:
the reference's text parser (src/common/parser.cpp through oracle/_ref/libref.so) vs our single-pass
text parser vs the .gcnbin cache, on the same files; all three must return identical arrays.

    python tests/validation/bench_loader.py [dataset ...]          (default: pubmed-syn reddit-mini)
"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import load_dataset, save_binary
from oracle.pyoracle import Ref

KEYS = ("g_indptr", "g_indices", "f_indptr", "f_indices", "f_val", "split", "label")


def main():
    names = sys.argv[1:] or ["pubmed-syn", "reddit-mini"]
    ref = Ref() if Ref.available() else None
    for name in names:
        ds = datagen.make_dataset(name)
        with tempfile.TemporaryDirectory() as td:
            root = os.path.join(td, "data")
            t0 = time.perf_counter(); datagen.write_text(ds, root); t_write = time.perf_counter() - t0
            size = sum(os.path.getsize(os.path.join(root, f)) for f in os.listdir(root))
            ours = load_dataset(root, name); t_ours = ours["_load_s"]
            save_binary(ours, os.path.join(root, name + ".gcnbin"))
            bin_size = os.path.getsize(os.path.join(root, name + ".gcnbin"))
            cached = load_dataset(root, name); t_bin = cached["_load_s"]
            line = f"{name}: text {size / 1e6:.1f} MB (written in {t_write:.1f} s), gcnbin {bin_size / 1e6:.1f} MB | ours text {t_ours:.2f} s | gcnbin {t_bin:.3f} s"
            for k in KEYS:
                assert np.array_equal(ours[k], cached[k]) and np.array_equal(ours[k], np.asarray(ds[k], ours[k].dtype)), k
            if ref:
                t0 = time.perf_counter(); theirs = ref.parse(td, name); t_ref = time.perf_counter() - t0
                for k in KEYS:
                    assert np.array_equal(ours[k], theirs[k]), k
                line += f" | reference parser {t_ref:.2f} s"
            print(line, flush=True)


if __name__ == "__main__":
    main()
