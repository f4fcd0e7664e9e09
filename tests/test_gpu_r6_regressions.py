"""Round-6 findings of the randomised cross-checks, kept as fixed cases.

1. tools/fuzz_components.py, seed 63, configuration 66 (256 x 192, template 41 x 11, sub-pixel refinement of the right
   keypoints on): cv::cornerSubPix leaves a match of a stripe at the image's upper edge 3.7 rows ABOVE the image, VALID;
   UndistorterRectifier::distortUnrectifyKeypoints (UndistorterRectifier.cpp:213-228) then reads
   map_x_.at<float>(round(y), round(x)) -- an unchecked read upstream (cv::Mat::at asserts in debug builds only), a read
   in front of the rectification map on the device: a GPU memory access fault when the map happened to begin a mapping
   (it took 66 configurations of allocation history to get there).  The library and the oracle now define the case as
   "the map entry of the nearest pixel"; this test holds the case itself (asserted on the oracle's output, so that it
   keeps testing what it says) and compares every output of StereoMatcher::sparseStereoReconstruction.

2. tools/fuzz_batched.py, seed 2, configuration 79 (12 streams of 752 x 480, klt_max_level 1, device-pointer steps with
   device_frames_persist 0): one run in three, a tracked keypoint of streams 8 - 11 differed from the oracle.  The
   context's own pyramid read back in place (kvfe_frontend_debug_pyramid) showed the level-0 copy with the first dword of
   some 16-byte groups replaced -- columns 192 - 255, 448 - 511, 704 - 751 of one row = lanes 12 - 15 of every row of 16
   lanes -- by [0, 0, c, b]: the v_perm_b32 that follows the copy's buffer_store_dwordx4 in pyr2_kernel and overwrites
   the store's first data register.  gfx950 reads the data of a store of more than 64 bits after issue; hipcc pads that
   hazard except for a store with an SGPR offset; a wave that issues back to back (the second group of 8 streams had
   only 4: its waves ran alone on their SIMDs) hits it.  The store now carries its row offset in the vector offset
   (k_rectify.hip, pyr2_strip), tools/check_store_data_hazard.py scans the compiled code of every kernel file
   (tests/test_host_logic.py), and the second test below repeats the launch that showed it.

3. tools/r6/gpu_guard.sh (the GPU suite with every device buffer of the library between unmapped guard ranges,
   KVFE_GUARD_ALLOC = 1 / 2 -- the pool has no GPU AddressSanitizer): kvfe_create for a 130 x 16 image faulted --
   rectify_box_kernel (context creation) read its four columns per lane without the test against the width the pack
   kernel beside it has, 16 bytes past the rectification map in the last row when the width is no multiple of 4.  (The
   boxes are only used for widths that are; with plain allocations the read landed in a neighbour.)  The last test runs
   context creation at such sizes, and the dense sequence that showed the allocator's own artefact
   (profiles/r6_analysis.md section 15), under the guard allocator in child processes."""
import os

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import frontend as F
from kimera_vio_amd import params as P
from kimera_vio_amd import synth, workloads

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case():
    w, h = 256, 192
    L, R = workloads.make_cameras(w, h)
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    p.stereo.templ_cols, p.stereo.templ_rows = 41, 11
    p.stereo.stripe_extra_rows = 0
    p.stereo.subpixel_refinement = 1
    p.stereo.min_point_dist = 0.3
    return w, h, L, R, p


def test_right_keypoint_refined_out_of_the_image_is_looked_up_at_the_nearest_pixel():
    w, h, L, R, p = _case()
    ctx = F.Context(L, R, p)
    try:
        oc = O.Camera(L, R)
        found = 0
        for seed in (979, 11, 12):
            st = synth.RigStream(L, R, seed=seed, rect_R1=np.array(ctx.rect.R1).reshape(3, 3))
            l0, r0 = st.frame(0)
            rng = np.random.RandomState(seed)
            # keypoints all over the image, and a band along the upper and lower edges (where a stripe is cut by the image)
            kps = np.stack([rng.uniform(0, w - 1, 150), rng.uniform(0, h - 1, 150)], 1).astype(np.float32)
            kps[:40, 1] = rng.uniform(0, 8, 40)
            kps[40:80, 1] = rng.uniform(h - 9, h - 1, 40)
            e = oc.sparse_stereo(l0, r0, kps, p.stereo)
            rx, rs = e["right_rect_xy"], e["right_status"]
            outside = (rs == 0) & ((np.rint(rx[:, 0]) < 0) | (np.rint(rx[:, 0]) > w - 1) |
                                   (np.rint(rx[:, 1]) < 0) | (np.rint(rx[:, 1]) > h - 1))
            found += int(outside.sum())
            g = ctx.sparse_stereo_reconstruction(l0, r0, kps)
            for k in ("left_rect_xy", "left_status", "right_rect_xy", "right_status", "depth", "keypoints_3d", "right_xy"):
                if k in e and k in g:
                    assert np.array_equal(g[k], e[k], equal_nan=True), (seed, k)
        assert found >= 1, "no VALID right keypoint outside the image in any of the scenes: the case is gone"
    finally:
        ctx.close()


def test_level0_copy_of_the_pyramid_launch_with_a_half_empty_stream_group():
    """finding 2: the launch itself (pyr2_kernel with the level-0 copy, 12 images = one full group of 8 streams and one
    with 4, whose waves run alone on their SIMDs), 150 times; the copy and every level against the oracle's cv::pyrDown.
    (In the front-end the build with the hazard failed one launch in ~20 -- tools/r6/gpu_pyr_probe.sh: 15 corrupted copies
    in 280 steps; this component call did not reproduce it on that build.  Kept as the plain check of the launch.)"""
    w, h = 752, 480
    L, R = workloads.make_cameras(w, h)
    for max_level in (1, 3):     # one-level launch (COPY, no second level) and the two-level launch of the shipped pyramid
        p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
        p.tracker.klt_max_level = max_level
        ctx = F.Context(L, R, p)
        try:
            rng = np.random.RandomState(5 + max_level)
            imgs = rng.randint(0, 256, (12, h, w)).astype(np.uint8)
            exp = []
            for s in range(12):
                lv, src = [], imgs[s]
                for _ in range(max_level):
                    src = O.pyr_down(src)
                    lv.append(src)
                exp.append(lv)
            for it in range(150):
                levels, cp = ctx.build_optical_flow_pyramid(imgs, with_level0_copy=True)
                if not np.array_equal(cp, imgs):
                    s_, ys, xs = np.nonzero(cp != imgs)
                    raise AssertionError("launch %d, klt_max_level %d: level-0 copy differs at %d bytes, stream %d row %d "
                                         "columns %s" % (it, max_level, len(ys), s_[0], ys[0], xs[:16].tolist()))
                for s in range(12):
                    assert len(levels[s]) == max_level
                    for l in range(max_level):
                        assert np.array_equal(levels[s][l], exp[s][l]), (it, max_level, s, l)
        finally:
            ctx.close()


def test_level0_copy_read_back_in_place_in_the_configuration_that_found_it():
    """finding 2 where it was found: tools/fuzz_batched.py seed 2, configuration 79, eight times in one process, with the
    context's own level-0 copy and pyramid read back after every step (FUZZ_PYR_PROBE: kvfe_frontend_debug_pyramid
    against the frame and the oracle's cv::pyrDown) besides the comparison of every output with the oracle.  The build with
    the hazard failed 10 - 11 of 30 / 10 of 40 such repeats (tools/r6/gpu_fb79.sh, gpu_pyr_probe.sh); neither the
    component call above nor a re-staging of the same steps with other scenes reproduced it there (tools/r6/
    gpu_store_hazard.sh) -- what makes a wave issue the store and the next instruction back to back is not understood
    beyond "it ran alone on its SIMD", which is why the static scan (tests/test_host_logic.py) is the check that counts."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_PYR_PROBE="1", FUZZ_REPEAT="8")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_batched.py"), "120", "2", "79"], env=env,
                       capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    m = re.search(r"configs failed: (\d+) of", out)
    assert m and int(m.group(1)) == 0 and "PYRAMID" not in out and "MISMATCH" not in out, out[-3000:]


@pytest.mark.parametrize("mode", ["1", "2"])
def test_context_creation_and_a_dense_sequence_under_guard_allocations(mode):
    """finding 3: KVFE_GUARD_ALLOC = 1 (buffers end where their mapping ends) / 2 (begin where it begins) in child processes
    (the switch is read once per process): context creation for widths that are no multiple of 4 and tiny heights, then
    eight dense calls with changing pair counts -- an out-of-bounds access of 16 bytes or more kills the child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KVFE_GUARD_ALLOC=mode)
    for w, h in ((130, 16), (323, 241), (750, 480)):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "r6", "guard_create_probe.py"), str(w), str(h)], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "done" in r.stdout, (mode, w, h, r.stdout[-500:], r.stderr[-1500:])
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "r6", "guard_dense_probe.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "done" in r.stdout, (mode, r.stdout[-500:], r.stderr[-1500:])
