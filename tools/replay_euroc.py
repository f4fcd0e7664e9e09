#!/usr/bin/env python3
"""Replay an EuRoC-layout dataset through the library end to end: the reference's chain

    EurocDataProvider -> StereoDataProviderModule -> [IMU rotation] -> StereoVisionImuFrontend::spinOnce

with every stage taken from libkvfe: PNG decode into the pinned staging slot (kvfe_png_decode_gray_batch),
left / right / IMU synchronisation (kvfe_stereo_sync_*), the gyro rotation since the last keyframe
(kvfe_imu_preintegrate_rotation -> kvfe_keyframe_R_cur_frame) and the staged front-end step on the GPU
(kvfe_frontend_step_staged).  One sequence = one stream; `--copies N` replays N copies of it as a batch.

    python tools/replay_euroc.py /data/V1_01_easy --final-k 400
    python tools/replay_euroc.py /root/reference/tests/data/MicroEurocDataset --dry-run     # no GPU: stops at the step

The host legs (provider, synchroniser, rotation, decode: --dry-run) are covered by tests/test_input_side.py; the whole
chain including the GPU step is driven by tests/test_gpu_replay_r3.py (`replay()` below, every step against the oracle).

The dataset's own cam0 / cam1 sensor.yaml give the calibration; the front-end parameters default to the shipped
params/Euroc/FrontendParams.yaml (tests/golden/params_euroc).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kimera_vio_amd import _abi as abi  # noqa: E402
from kimera_vio_amd import dataprovider as dp  # noqa: E402
from kimera_vio_amd import params as P  # noqa: E402


def packets(dataset, initial_k, final_k):
    """(frame index, timestamp, left file bytes, right file bytes, imu stamps, imu acc_gyr) for every synchronised pair"""
    prov = dp.EurocDataProvider(dataset, initial_k, final_k)
    sync = dp.StereoDataProviderModule(-1)
    files = {}

    # the provider's callbacks get decoded images upstream; here the FILES travel to the staging slot and are
    # decoded there, so the callbacks only register the frames (one read of each file, no intermediate image)
    def reg(side):
        def cb(k, t):
            name = prov.getLeftImgName(k) if side == 0 else prov.getRightImgName(k)
            with open(name, "rb") as f:
                files[(side, k)] = f.read()
            (sync.fillLeftFrameQueue if side == 0 else sync.fillRightFrameQueue)(t, k)
        return cb
    left_cb, right_cb = reg(0), reg(1)
    for t, ag in zip(prov.imu_timestamps, prov.imu_acc_gyr):
        sync.fillImuQueue(int(t), ag)
    for k in range(prov.initial_k, prov.final_k):
        ln, rn = prov.getLeftImgName(k), prov.getRightImgName(k)
        if not (ln and rn and os.path.exists(ln) and os.path.exists(rn)):
            continue                                          # "Missing left/right stereo pair"
        t = prov.timestampAtFrame(k)
        left_cb(k, t)
        right_cb(k, t)
        while True:
            pk = sync.getInputPacket()
            if pk is not None:
                yield pk.left_tag, pk.timestamp, files.pop((0, pk.left_tag)), files.pop((1, pk.right_tag)), \
                    pk.imu_stamps, pk.imu_accgyrs
                continue
            if sync.last_action in (abi.SYNC_EMPTY, abi.SYNC_WAIT_IMU):
                break
        for key in [key for key in files if key[1] < k - 4]:   # frames the synchroniser dropped
            del files[key]


def replay(dataset, L, R, ctx, copies=1, threads=0, gyro_bias=(0.0, 0.0, 0.0), initial_k=0, final_k=1 << 30,
           host=None, read_back=True):
    """The chain itself, one dict per synchronised stereo pair:
        {k, ts, imu_n, deltaRij (since the last keyframe, before this step's reset), Rk = keyframe_R_cur_frame,
         left, right (the decoded frames of this step: staging slot views, valid until the slot is reused),
         out (kvfe_frontend_get_output of stream 0, when a context is given)}
    ctx = None: everything but the GPU step (frames are decoded into `host`)."""
    from kimera_vio_amd import frontend as F
    B = copies
    rect_R1 = np.array(ctx.rect.R1 if ctx is not None else F.compute_rectification(L, R).R1).reshape(3, 3)
    body_R_camLrect = np.array(L.body_pose_cam).reshape(4, 4)[:3, :3] @ rect_R1.T   # getBodyPoseLeftCamRect().rotation()
    deltaRij = np.eye(3)
    slot = 0
    if ctx is None and host is None:
        host = np.empty((2 * B, L.height, L.width), np.uint8)
    for k, ts, lf, rf, imu_t, imu_ag in packets(dataset, initial_k, final_k):
        deltaRij = dp.preintegrate_rotation(imu_t, imu_ag, gyro_bias, deltaRij)
        Rk = dp.keyframe_R_cur_frame(body_R_camLrect, deltaRij)
        td = time.perf_counter()
        if ctx is None:
            dp.decode_png_gray_batch([lf] * B + [rf] * B, host, threads)
            left, right = host[:B], host[B:]
        else:
            ctx.staging_wait(slot)
            left, right = ctx.staging_buffers(slot)
            dp.decode_png_gray_batch([lf] * B, left, threads)
            dp.decode_png_gray_batch([rf] * B, right, threads)
        rec = dict(k=k, ts=ts, imu_n=int(imu_t.size), deltaRij=deltaRij.copy(), Rk=Rk, left=left, right=right,
                   decode_s=time.perf_counter() - td, out=None)
        if ctx is not None:
            ctx.step_staged(slot, ctx.make_inputs([ts] * B, [Rk] * B, [0] * B))
            if read_back:
                rec["out"] = ctx.get_output(0)                # (a replay reads every frame back; a pipeline need not)
                if rec["out"]["is_keyframe"]:
                    deltaRij = np.eye(3)                      # ImuFrontend::resetIntegrationWithCachedBias
            slot = (slot + 1) % 3
        yield rec


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dataset")
    ap.add_argument("--initial-k", type=int, default=0)
    ap.add_argument("--final-k", type=int, default=1 << 30)
    ap.add_argument("--frontend-params", default=os.path.join(ROOT, "tests", "golden", "params_euroc", "FrontendParams.yaml"))
    ap.add_argument("--copies", type=int, default=1, help="replay this many copies of the sequence as one batch")
    ap.add_argument("--threads", type=int, default=0, help="decode threads (0: one per file up to the host's cores)")
    ap.add_argument("--gyro-bias", type=float, nargs=3, default=(0.0, 0.0, 0.0))
    ap.add_argument("--dry-run", action="store_true", help="everything but the GPU step (no device needed)")
    a = ap.parse_args()

    L = P.load_camera_params(os.path.join(a.dataset, "mav0", "cam0", "sensor.yaml"))
    R = P.load_camera_params(os.path.join(a.dataset, "mav0", "cam1", "sensor.yaml"))
    p = P.load_frontend_params(a.frontend_params)
    B = a.copies
    ctx = None
    if not a.dry_run:
        from kimera_vio_amd import frontend as F
        ctx = F.Context(L, R, p, batch=B)
    n, t_decode, t0 = 0, 0.0, time.perf_counter()
    for rec in replay(a.dataset, L, R, ctx, B, a.threads, a.gyro_bias, a.initial_k, a.final_k):
        dR = rec["deltaRij"]
        line = (f"frame {rec['k']} t={rec['ts']} imu={rec['imu_n']} "
                f"|dR|={np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))):.3f}deg")
        out = rec["out"]
        if out is not None:
            line += (f" keypoints={out['n_keypoints']} keyframe={int(out['is_keyframe'])} "
                     f"mono={out['tracking_status_mono']} stereo={out['tracking_status_stereo']}")
        else:
            line += f" left_mean={rec['left'][0].mean():.2f}"
        print(line)
        t_decode += rec["decode_s"]
        n += 1
    dt = time.perf_counter() - t0
    print(f"{n} pairs x {B} stream(s) in {dt:.2f} s: {n * B / dt:.1f} pairs/s end to end, decode {1e3 * t_decode / max(n, 1):.2f} ms per "
          f"batch of {2 * B} files" + ("" if ctx is not None else " (dry run: no front-end step)"))


if __name__ == "__main__":
    main()
