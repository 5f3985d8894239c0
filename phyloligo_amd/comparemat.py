#!/usr/bin/env python3
"""Mirror of /root/reference/phylopackage/bin/phyloligo_comparemat.py: read two distance matrices in any of the three
formats phyloligo writes and say whether they agree (numpy.allclose, atol 1e-3, :44).

    python -m phyloligo_amd.comparemat --mat1 a.mat --format1 numpy --mat2 b.f32 --format2 memmap

Formats (reference :7-24): `numpy` = the text .mat (numpy.loadtxt), `memmap` = headerless float32, N = sqrt(length)
("weird shape" -> exit 1), `h5py` = HDF5 file with the dataset "distances" - read here through libhdf5 (phyloligo_amd/hdf5.py),
since h5py itself is not a dependency.  Same option names, same output lines, exit status 0.  Host only: no GPU.
"""
import argparse
import sys

import numpy as np


def read_numpy(path):
    return np.loadtxt(path)


def read_h5py(path):
    from . import hdf5
    return hdf5.read_f32_dataset(path, "distances")


def read_memmap(path):
    matrix = np.memmap(path, dtype=np.float32, mode="r")
    s = matrix.shape[0]
    n = np.sqrt(s)
    if str(n).split(".")[1] != "0":
        print("Error, weird shape for matrix {}".format(path), file=sys.stderr)
        sys.exit(1)
    return matrix.reshape((int(n), int(n)))


format2fn = {"numpy": read_numpy, "h5py": read_h5py, "memmap": read_memmap}


def main(argv=None):
    parser = argparse.ArgumentParser(prog="phyloligo_comparemat.py")
    parser.add_argument("--mat1", action="store", dest="matrix1")
    parser.add_argument("--format1", action="store", dest="format1", choices=["numpy", "memmap", "h5py"])
    parser.add_argument("--mat2", action="store", dest="matrix2")
    parser.add_argument("--format2", action="store", dest="format2", choices=["numpy", "memmap", "h5py"])
    params = parser.parse_args(argv)
    mat1 = format2fn[params.format1](params.matrix1)
    mat2 = format2fn[params.format2](params.matrix2)
    print("matrix {}, shape: {}".format(params.matrix1, mat1.shape))
    print("matrix {}, shape: {}".format(params.matrix2, mat2.shape))
    print("Identical matrices?:", np.allclose(mat1, mat2, atol=1e-3))
    print()
    print(mat1)
    print()
    print(mat2)
    return 0


if __name__ == "__main__":
    main()
    sys.exit(0)
