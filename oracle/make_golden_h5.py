"""HDF5 hand-off files written by the REAL h5py, exactly as utils.py:92-99 writes them (``hf.create_dataset('data', data=im,
compression='lzf')``).  TEST INFRASTRUCTURE ONLY.

The image's main interpreter has no h5py, but its Anaconda tree does (h5py 3.3.0 on HDF5 1.10.6, Python 3.9), so this script runs under

    /opt/conda/bin/python3.9 oracle/make_golden_h5.py          # writes tests/golden/h5/*.h5 + *.npy

and imports nothing of this repository.  Fixtures (small on purpose): the cascade hand-off's shapes in miniature -- the packed SG parameters
``imenv_*_0.h5 [7*SGNum = 84, R, C]`` (outputBRDFLight.py:289-301), ``imdiffuse / imspecular_*_0.h5 [3, R, C]`` (:277-287) --, a compressible
and an incompressible array (h5py stores a chunk raw when LZF does not shrink it: the filter is registered optional), a multi-chunk array,
and an UNCOMPRESSED dataset.  Each ``.h5`` has the same values as ``.npy``; tests/test_h5_handoff.py reads the former with the product's
reader and compares bit for bit."""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "h5")


def write_like_the_reference(path, im, compression="lzf"):
    hf = h5py.File(path, "w")                                   # utils.py:97
    hf.create_dataset("data", data=im, compression=compression)  # utils.py:98
    hf.close()


def main():
    os.makedirs(OUT, exist_ok=True)
    rs = np.random.RandomState(20259)
    R, C, K = 12, 16, 12
    smooth = np.linspace(0.0, 1.0, R * C, dtype=np.float32).reshape(1, R, C)
    cases = {
        "imenv_handoff": np.tanh(rs.standard_normal((7 * K, R, C))).astype(np.float32),              # raw SG parameters: hardly compressible
        "imdiffuse_handoff": (smooth * np.array([0.2, 0.5, 0.9], dtype=np.float32).reshape(3, 1, 1)).astype(np.float32),
        "imspecular_zeros": np.zeros((3, R, C), dtype=np.float32),                                   # one long run
        "quantised_multichunk": (rs.randint(0, 4, size=(84, 30, 40)).astype(np.float32) * 0.25),      # several chunks, compressible
        "noise_incompressible": rs.standard_normal((3, 40, 50)).astype(np.float32),
        "one_value": np.array([[[3.25]]], dtype=np.float32),
    }
    for name, arr in cases.items():
        write_like_the_reference(os.path.join(OUT, name + ".h5"), arr)
        np.save(os.path.join(OUT, name + ".npy"), arr)
        with h5py.File(os.path.join(OUT, name + ".h5"), "r") as hf:
            d = hf["data"]
            print(f"{name:24s} shape {d.shape} chunks {d.chunks} compression {d.compression} file {os.path.getsize(os.path.join(OUT, name + '.h5'))} B")
    write_like_the_reference(os.path.join(OUT, "uncompressed.h5"), cases["imdiffuse_handoff"], compression=None)
    np.save(os.path.join(OUT, "uncompressed.npy"), cases["imdiffuse_handoff"])
    print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
