"""INTEGRATION.md section 1 for real: the reference-side Reduction policy include/small_gicp/registration/reduction_hip.hpp is
compiled together with the UNMODIFIED reference headers (Registration<Factor, ParallelReductionHIP>, registration/registration.hpp:
17-54) and linked against libsmall_gicp_amd.so by oracle/ref/Makefile -> oracle/_ref/test_reduction_hip.  The binary runs the
reference's own Registration<Factor, ParallelReductionOMP> and the HIP policy side by side on config C1 and checks pose (1e-4),
iteration count, num_inliers (through optimizer.hpp:146, i.e. the host factors the policy filled), cached uploads and the refill-in-
place case.  /root/reference exists only in the build container: the GPU box runs the prebuilt binary."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "oracle", "_ref", "test_reduction_hip")


def test_policy_builds_from_the_unmodified_reference_headers():
    """Where the reference tree is mounted (the build container), the policy + test program must compile."""
    if not os.path.isdir("/root/reference/include/small_gicp"):
        pytest.skip("no reference tree here (GPU box): the binary was built in the build container")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref")], stdout=subprocess.DEVNULL)
    assert os.path.exists(BIN)


def _syntax_only(tmp_path, body):
    src = tmp_path / "vm.cpp"
    src.write_text(
        "#include <small_gicp/ann/flat_container.hpp>\n#include <small_gicp/ann/gaussian_voxelmap.hpp>\n#include <small_gicp/points/point_cloud.hpp>\n#include <small_gicp/registration/registration.hpp>\n"
        "#include <small_gicp/registration/reduction_hip.hpp>\nusing namespace small_gicp;\n" + body)
    ref = os.path.join(ROOT, "oracle", "ref")
    return subprocess.run(["g++", "-std=c++17", "-fopenmp", "-w", "-fsyntax-only", "-I" + os.path.join(ref, "eigen_shim"), "-I/root/reference/include", "-I" + os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)


def test_policy_takes_the_reference_voxel_maps_as_targets(tmp_path):
    """registration_helper.cpp:125-137 runs VGICP as Registration<GICPFactor, Reduction> with a GaussianVoxelMap as target and tree, the model-based
    odometry (odometry_benchmark_small_gicp_model_omp.cpp) does the same with an IncrementalVoxelMap<FlatContainerCov>: the HIP policy takes
    both (tests/cpp/test_reduction_hip.cpp runs them on the GPU).  A voxel map of contents it does not know must refuse to compile —
    traits::point of a voxel map takes packed indices, the generic upload would read garbage — with a message that says why."""
    if not os.path.isdir("/root/reference/include/small_gicp"):
        pytest.skip("no reference tree here (GPU box)")
    for decl in ("GaussianVoxelMap vm(0.5);", "IncrementalVoxelMap<FlatContainerCov> vm(0.5);", "IncrementalVoxelMap<FlatContainerPoints> vm(0.5);"):
        ok = _syntax_only(tmp_path, "int main() { %s PointCloud src; Registration<GICPFactor, ParallelReductionHIP> reg; auto r = reg.align(vm, src, vm); return (int)r.iterations; }\n" % decl)
        assert ok.returncode == 0, (decl, ok.stderr[-2000:])
    bad = _syntax_only(tmp_path, "struct Blob { struct Setting {}; size_t size() const { return 0; } };\n"
                                 "int main() { IncrementalVoxelMap<Blob> vm(0.5); PointCloud src; ParallelReductionHIP red; red.bind(vm, src, Eigen::Isometry3d::Identity()); return 0; }\n")
    assert bad.returncode != 0 and "only GaussianVoxelMap" in bad.stderr, bad.stderr[-2000:]


def test_policy_host_side_pieces():
    """tests/cpp/test_policy_host.cpp: repacking of clouds and of the reference's voxel maps for the C ABI, the content hash (any single entry
    changed is noticed), the reference's set_search_offsets(27) quirk the device's 27-voxel order follows.  No GPU: runs wherever the binary is."""
    host = os.path.join(ROOT, "oracle", "_ref", "test_policy_host")
    if not os.path.exists(host):
        pytest.skip("oracle/_ref/test_policy_host is not built (make -C oracle/ref where /root/reference is mounted)")
    p = subprocess.run([host], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "DONE failures=0" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.gpu
def test_registration_with_the_hip_reduction_policy(tmp_path):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/test_reduction_hip did not travel with the repository (build it with `make -C oracle/ref` where /root/reference is mounted)")
    d = np.load(os.path.join(GOLDEN, "c1_points.npz"))
    for name in ("target", "source"):
        np.ascontiguousarray(d[name][:, :3], dtype="<f4").tofile(tmp_path / (name + ".bin"))
    p = subprocess.run([BIN, str(tmp_path / "target.bin"), str(tmp_path / "source.bin")], capture_output=True, text=True, timeout=600)
    cases = [json.loads(ln[5:]) for ln in p.stdout.splitlines() if ln.startswith("CASE ")]
    assert p.returncode == 0 and len(cases) >= 40 and all(c["ok"] for c in cases), p.stdout[-3000:] + p.stderr[-2000:]
    for ln in p.stdout.splitlines():
        if ln.startswith("RATE "):
            print("policy rate:", ln[5:])
    rates = [json.loads(ln[5:]) for ln in p.stdout.splitlines() if ln.startswith("RATE ")]
    assert len(rates) == 5 and all(r["hip_policy_iterations_per_s"] > 0 and r["uploads"] == 2 for r in rates)  # two uploads: target and source, once
    gicp = cases[0]
    assert gicp["num_inliers"][0] == gicp["reduction_num_inliers"] > 5000  # RegistrationResult::num_inliers is right without patching the optimizer


@pytest.mark.gpu
def test_reference_helper_api_served_by_the_hip_implementation(tmp_path, c1_gold):
    """integration/registration_helper_hip.cpp defines the functions registration/registration_helper.hpp declares — the reference's compiled
    helper library (src/small_gicp/registration/registration_helper.cpp) with the MI355X path behind the same symbols.  tests/cpp/
    test_helper_hip.cpp calls them the way the reference's examples do (the one-call Eigen interface, preprocess_points + align for every
    registration type, create_gaussian_voxelmap + VGICP); the poses must equal the goldens of config C1 within the 1e-4 m / 1e-4 rad bar."""
    from conftest import pose_error

    binary = os.path.join(ROOT, "oracle", "_ref", "test_helper_hip")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/test_helper_hip did not travel with the repository (make -C oracle/ref where /root/reference is mounted)")
    d = np.load(os.path.join(GOLDEN, "c1_points.npz"))
    for name in ("target", "source"):
        np.ascontiguousarray(d[name][:, :3], dtype="<f4").tofile(tmp_path / (name + ".bin"))
    p = subprocess.run([binary, str(tmp_path / "target.bin"), str(tmp_path / "source.bin")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = {r["name"]: r for r in (json.loads(ln[7:]) for ln in p.stdout.splitlines() if ln.startswith("RESULT "))}
    assert set(res) == {"points_GICP", "points_VGICP", "ICP", "PLANE_ICP", "GICP", "VGICP"}, p.stdout[-2000:]
    pre = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("PREPROCESSED ")][0][13:])
    assert [pre["target"], pre["source"]] == c1_gold["downsampled_sizes"] and pre["tree"] == pre["target"]
    edited = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("EDITED ")][0][7:])
    assert abs(edited["dx"] + 0.05) < 2e-3, edited  # a cloud edited after preprocess_points is NOT served from its device twin
    timing = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("TIMING ")][0][7:])
    print("helper timing:", timing)
    tree = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("TREE ")][0][5:])
    assert tree["found"] == 1 and tree["index"] == 0 and tree["sq_dist"] < 1e-12  # the returned KdTree is a working reference tree
    for name, gold in (("points_GICP", "GICP"), ("GICP", "GICP"), ("ICP", "ICP"), ("PLANE_ICP", "PLANE_ICP"), ("points_VGICP", "VGICP"), ("VGICP", "VGICP")):
        r, g = res[name], c1_gold["cases"][gold]
        dt, dr = pose_error(np.array(r["T"]).reshape(4, 4).T, np.array(g["T"]))
        assert dt < 1e-4 and dr < 1e-4 and bool(r["converged"]) == g["converged"], (name, dt, dr)
        assert abs(r["iterations"] - g["iterations"]) <= 1 and abs(r["num_inliers"] - g["num_inliers"]) <= 3, (name, r["iterations"], g["iterations"], r["num_inliers"], g["num_inliers"])
        print("helper %s: dt %.2e dr %.2e iterations %d inliers %d (golden %d)" % (name, dt, dr, r["iterations"], r["num_inliers"], g["num_inliers"]))
