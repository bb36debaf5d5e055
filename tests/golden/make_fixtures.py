#!/usr/bin/env python3
"""Convert the reference's own test data files into compact float32 fixtures.

Run in the dev container only (reads /root/reference, which does not exist on the
GPU box).  Outputs are DATA (point coordinates / normals / expected transforms),
not reference source:

  car_cloud400.npy  (24989, 6) float32  x y z nx ny nz   <- libpointmatcher/examples/data/car_cloud400.csv
  car_cloud401.npy  (25193, 3) float32  x y z            <- libpointmatcher/examples/data/car_cloud401.csv
  validT3d.npy      (4, 4) float32  expected transform of utest.cpp:356-360 (tolerance 0.1 / 0.1 rad, utest.h:65-86)
  cloud00000.npy    (24989, 3) float32  x y z            <- libpointmatcher/examples/data/cloud.00000.vtk (POINTS block)
  cloud00001.npy    (25193, 3) float32  x y z            <- libpointmatcher/examples/data/cloud.00001.vtk (POINTS block)
  icp_data_surface_normal_p2pl_ref_trans.npy (4, 4) float64
                    the 16 numbers of icp_data/defaultOrientNormalsDataPointsFilter.ref_trans -- byte-identical in
                    defaultObservationDirectionDataPointsFilter.ref_trans and defaultSimpleSensorNoiseDataPointsFilter.ref_trans
                    (three chains that differ only in descriptor-adding reading filters): the reference's golden for
                    "SurfaceNormalDataPointsFilter knn 10 on the reference -> KDTree knn 1 eps 0 -> Trimmed 0.75 ->
                    PointToPlane, Counter 40, Differential 0.001 / 0.01 / 4" on cloud.00001 -> cloud.00000
                    (utest.cpp:81-161, criterion :146-159)

The CSV text is parsed with float64 and rounded once to float32, which is what
libpointmatcher's loader does when instantiated with T=float (IO.cpp loadCSV -> T).
"""
import os
import numpy as np

REF = "/root/reference/libpointmatcher/examples/data"
OUT = os.path.dirname(os.path.abspath(__file__))


def vtk_points(path):
    """POINTS block of an ASCII legacy-VTK POLYDATA file (IO.cpp loadVTK reads the same numbers into T=float)."""
    tok = open(path).read().split()
    i = tok.index("POINTS")
    n = int(tok[i + 1])
    vals = np.array(tok[i + 3:i + 3 + 3 * n], dtype=np.float64)
    return vals.reshape(n, 3)


def main():
    a = np.loadtxt(os.path.join(REF, "car_cloud400.csv"), delimiter=",", skiprows=1, dtype=np.float64)
    assert a.shape == (24989, 6), a.shape
    np.save(os.path.join(OUT, "car_cloud400.npy"), a.astype(np.float32))
    b = np.loadtxt(os.path.join(REF, "car_cloud401.csv"), dtype=np.float64)
    assert b.shape == (25193, 3), b.shape
    np.save(os.path.join(OUT, "car_cloud401.npy"), b.astype(np.float32))
    # utest/utest.cpp:356-360 (values typed from the test's known answer)
    validT3d = np.array([[0.982304, 0.166685, -0.0854066, 0.0446816],
                         [-0.150189, 0.973488, 0.172524, 0.191998],
                         [0.111899, -0.156644, 0.981296, -0.0356313],
                         [0, 0, 0, 1]], dtype=np.float32)
    np.save(os.path.join(OUT, "validT3d.npy"), validT3d)
    for name in ("cloud.00000.vtk", "cloud.00001.vtk"):
        pts = vtk_points(os.path.join(REF, name))
        np.save(os.path.join(OUT, name.replace(".", "").replace("vtk", "") + ".npy"), pts.astype(np.float32))
    icp_data = os.path.join(REF, "icp_data")
    golden = [np.loadtxt(os.path.join(icp_data, f + ".ref_trans"), dtype=np.float64) for f in
              ("defaultOrientNormalsDataPointsFilter", "defaultObservationDirectionDataPointsFilter",
               "defaultSimpleSensorNoiseDataPointsFilter")]
    assert all(g.shape == (4, 4) and np.array_equal(g, golden[0]) for g in golden)
    np.save(os.path.join(OUT, "icp_data_surface_normal_p2pl_ref_trans.npy"), golden[0])
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
