"""steps enqueued back to back with NO synchronisation in between -- the previous step's tail (stereo matching
of its new corners + finalisation) runs on the library's side stream while the next step is already being enqueued, so
every buffer the next step's entry point touches before do_step must be ordered after that tail:
  * equalize_image = 1 + kvfe_frontend_step_host uploads the raw frames into the rectified buffers as scratch,
  * kvfe_config.copy_inputs copies the per-stream inputs into the single device copy the finalisation reads,
  * a caller-owned hip_stream must cover the whole step (work enqueued on it afterwards runs after the tail).
Each case replays MicroEuroc frames with every frame a keyframe (the tail always has work) and compares the LAST
step's full output with the oracle: any corruption of an earlier keyframe's right keypoints / depths changes the stereo
outlier rejection and the measurements of the following ones."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import frontend as F
from parity_util import assert_step_equal
from test_gpu_parity import _euroc_ransac_params, _kf_rotations, euroc_cams, ocam, seq  # noqa: F401

pytestmark = pytest.mark.gpu


def _replay(seq, ocam, equalize, n=8, B=2, hip_stream=None, sync_every=0, force=None, features=200,
            identity_streams=(), **ctx_kw):
    seq = dict(seq)
    seq["camR"] = _kf_rotations(seq["body_R"], ocam)
    L, R = euroc_cams()
    p = _euroc_ransac_params(max_features_per_frame=features)
    p.stereo.equalize_image = equalize
    fe = [O.Frontend(L, R, p) for _ in range(B)]
    c = F.Context(L, R, p, batch=B, hip_stream=hip_stream, **ctx_kw)
    try:
        exp, hist = None, []
        lkf = [None] * B   # frame index of each stream's last keyframe (from the oracle: nothing is read back from the GPU)
        for i in range(n):
            pp = lambda j: (j % 16) if (j % 16) < 9 else 16 - (j % 16)   # ping-pong over the 9 frames
            idx = ([i, 8 - i] + [pp(i + 3 * s) for s in range(2, B)])[:B]
            # keyframe_R_cur_frame: rotation from the stream's last keyframe to this frame
            Rs = [np.eye(3) if (lkf[s] is None or s in identity_streams) else seq["camR"][lkf[s]].T @ seq["camR"][idx[s]]
                  for s in range(B)]
            ts = [int(seq["ts"][i])] * B
            fk = [1] * B if force is None else [int(force[(i + s) % len(force)]) for s in range(B)]
            lefts = np.stack([seq["lefts"][j] for j in idx])
            rights = np.stack([seq["rights"][j] for j in idx])
            c.step_host(lefts, rights, c.make_inputs(ts, Rs, fk))      # enqueue only
            if sync_every and (i + 1) % sync_every == 0:
                c.synchronize()
            exp = [fe[s].process(lefts[s], rights[s], ts[s], Rs[s], bool(fk[s])) for s in range(B)]
            hist.append(exp)
            for s in range(B):
                if exp[s]["is_keyframe"]:
                    lkf[s] = idx[s]
        for s in range(B):
            assert_step_equal(c.get_output(s), exp[s], ("last", s))
            # the output ring: the records of the two steps before the last one are still there (kvfe_frontend_get_output_at)
            for back in (1, 2):
                if len(hist) > back:
                    assert_step_equal(c.get_output(s, steps_back=back), hist[-1 - back][s], ("back", back, s))
            if force is None:
                assert exp[s]["is_keyframe"] and exp[s]["n_measurements"] > min(50, features // 2)
        return exp
    finally:
        c.close()


@pytest.mark.parametrize("equalize", [1, 0])
def test_pipelined_host_steps_no_sync(seq, ocam, equalize):
    _replay(seq, ocam, equalize)


def test_pipelined_host_steps_user_stream(seq, ocam):
    """kvfe_config.hip_stream: the step is stream-ordered on the caller's stream as a whole"""
    import torch
    st = torch.cuda.Stream()
    _replay(seq, ocam, 1, hip_stream=st.cuda_stream)
    st.synchronize()


def test_pipelined_host_steps_copy_inputs(seq, ocam):
    """kvfe_config.copy_inputs = 1: the per-stream inputs travel by a H2D copy into the single device copy the
    finalisation reads (the previous step's tail must be joined first)"""
    _replay(seq, ocam, 1, copy_inputs=1)


def test_pipelined_host_steps_single_hip_stream(seq, ocam):
    """kvfe_config.single_hip_stream = 1: no side stream, no output stream -- every kernel in order on one stream"""
    _replay(seq, ocam, 0, single_hip_stream=1)
    _replay(seq, ocam, 1, B=6, single_hip_stream=1)


def test_pipelined_host_steps_many_streams_output_stream(seq, ocam):
    """more than four streams: the output records go through the device staging buffer and the output stream"""
    _replay(seq, ocam, 0, B=6)


def test_output_transfer_smaller_than_the_records():
    """round 4: the step's records lie back to back behind an offset table and travel in one DMA transfer whose size the
    host has to choose before the device knows the counts (the last completed step's need + a margin).  An access that
    finds a record beyond the transferred bytes fetches the rest from the staging buffer.  KVFE_OUT_TRANSFER_BYTES (read
    at kvfe_create, hence the sub-process) forces transfers of 4 KB -- less than one record -- so that every access of the
    many-stream replay (last step and the two before it, every stream) goes through that path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_pipelined_r3.py") + "::test_pipelined_host_steps_many_streams_output_stream",
                        os.path.join(root, "tests", "test_gpu_bench_configs.py") + "::test_c3_headline_64_streams_600_features"],
                       env=dict(os.environ, KVFE_OUT_TRANSFER_BYTES="4096"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_mixed_identity_and_gyro_rotations_in_one_batch(seq, ocam):
    """the host skips the launch of the 3-point (Arun) stereo rejection when no stream of the batch needs
    it, with the SAME predicate the kernels use (rot_is_identity, kvfe_dev.hpp).  Streams 0 and 2 never get a gyro
    rotation (identity: mono 5-point / stereo 3-point problems apply), streams 1 and 3 do (2-point / 1-point): both
    kinds in one launch, every stream equal to its oracle."""
    _replay(seq, ocam, 0, B=4, identity_streams=(0, 2))
    _replay(seq, ocam, 0, B=2, identity_streams=(0, 1))
    _replay(seq, ocam, 0, B=2)
