#!/usr/bin/env python3
"""Convert the reference's HDF5 label files (.../lbl/.../name.h5 with start_frame_lbl, end_frame_lbl, lbl_cat, foc;
nn/audio_tasks.py:335-343) to .npz files with the same four arrays next to them, for machines without h5py
(animal2vec_amd/data.py reads either).  Run where h5py is installed:

    python tools/convert_labels.py /data/MeerKAT/lbl
"""
import os
import sys

import numpy as np


def main():
    try:
        import h5py
    except ImportError:
        sys.exit("h5py is required to read the .h5 label files")
    root = sys.argv[1]
    n = 0
    for d, _, files in os.walk(root):
        for f in files:
            if not f.endswith(".h5"):
                continue
            src = os.path.join(d, f)
            with h5py.File(src, "r") as h:
                arrays = {k: np.asarray(h[k]) for k in ("start_frame_lbl", "end_frame_lbl", "lbl_cat", "foc") if k in h}
            np.savez(src[:-3] + ".npz", **arrays)
            n += 1
    print(f"converted {n} label files under {root}")


if __name__ == "__main__":
    main()
