"""BASELINE configs[0] / [1] plumbing END TO END on the GPU: an EuRoC-layout clip on disk goes through

    EurocDataProvider -> StereoDataProviderModule -> kvfe_imu_preintegrate_rotation -> kvfe_keyframe_R_cur_frame
    -> PNG decode into the pinned staging slot -> kvfe_frontend_step_staged -> kvfe_frontend_get_output

(`tools/replay_euroc.replay`, every stage from libkvfe; the reference chain is EurocDataProvider.cpp:146-195 ->
StereoDataProviderModule.cpp:35-91 -> StereoVisionImuFrontend::spinOnce) and every step is compared, tolerance 0, with the
oracle front-end fed the same packets from its OWN restatement of the chain: PIL-decoded frames, oracle/input_side.py's
synchroniser and gyro preintegration, numpy for camLrect_R_body . deltaRij . body_R_camLrect.

The clip is MicroEuroc frames 10..18 with the IMU rows around them (tests/golden/micro_euroc_f10_18.npz,
micro_euroc_imu_f10_18.csv -- committed fixtures, the GPU box has no /root/reference), written to a temporary directory
in the EuRoC layout."""
import importlib.util
import os
import shutil
import sys

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O  # noqa: E402
from kimera_vio_amd import frontend as F  # noqa: E402
from kimera_vio_amd import params as P  # noqa: E402
from oracle import input_side as ora  # noqa: E402
from parity_util import assert_step_equal  # noqa: E402

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _write_clip(root):
    z = np.load(os.path.join(G, "micro_euroc_f10_18.npz"))
    ts = [int(t) for t in z["timestamps"]]
    for cam, frames, yaml in (("cam0", z["lefts"], "sensorLeft.yaml"), ("cam1", z["rights"], "sensorRight.yaml")):
        d = os.path.join(root, "mav0", cam, "data")
        os.makedirs(d)
        lines = ["#timestamp [ns],filename"]
        for t, img in zip(ts, frames):
            Image.fromarray(img).save(os.path.join(d, f"{t}.png"))
            lines.append(f"{t},{t}.png")
        with open(os.path.join(root, "mav0", cam, "data.csv"), "w") as f:
            f.write("\n".join(lines) + "\n")
        shutil.copy(os.path.join(G, yaml), os.path.join(root, "mav0", cam, "sensor.yaml"))
    os.makedirs(os.path.join(root, "mav0", "imu0"))
    shutil.copy(os.path.join(G, "micro_euroc_imu_f10_18.csv"), os.path.join(root, "mav0", "imu0", "data.csv"))
    return z, ts


def _oracle_packets(root, ts):
    """the oracle's own synchroniser on the clip's index and IMU file: [(timestamp, frame k, imu stamps, imu rows)]"""
    prov = ora.StereoProvider(0)
    for ln in open(os.path.join(root, "mav0", "imu0", "data.csv")):
        if ln.startswith("#"):
            continue
        v = ln.strip().split(",")
        gyro, acc = [float(x) for x in v[1:4]], [float(x) for x in v[4:7]]
        prov.imu.add(int(v[0]), acc + gyro)           # the reference's ImuAccGyr: acceleration first
    out = []
    for k, t in enumerate(ts):
        prov.left.append((t, k))
        prov.right.append((t, k))
        while True:
            st, pk = prov.spin()
            if st == prov.PACKET:
                out.append((pk[0], pk[1], pk[3], pk[4]))
            if st in (prov.EMPTY, prov.WAIT_IMU):
                break
    return out


def test_euroc_clip_replayed_end_to_end_equals_oracle(tmp_path):
    spec = importlib.util.spec_from_file_location("replay_euroc", os.path.join(os.path.dirname(__file__), "..", "tools",
                                                                               "replay_euroc.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    root = str(tmp_path)
    z, ts = _write_clip(root)
    L = P.load_camera_params(os.path.join(root, "mav0", "cam0", "sensor.yaml"))
    R = P.load_camera_params(os.path.join(root, "mav0", "cam1", "sensor.yaml"))
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"))     # as shipped: useRANSAC 1
    B = 2
    ctx = F.Context(L, R, p, batch=B)
    fe = O.Frontend(L, R, p)
    opk = _oracle_packets(root, ts)
    assert [k for _, k, _, _ in opk] == list(range(1, 9))        # frame 0 only sets the previous stamp
    body_R_cam = np.array(L.body_pose_cam).reshape(4, 4)[:3, :3] @ np.array(ctx.rect.R1).reshape(3, 3).T
    dR = np.eye(3)
    n_kf = 0
    try:
        n_steps = 0
        # (lazily: a step's staging-slot views are only valid until the slot is reused three steps later)
        for i, (rec, (ot, ok, ostamps, orows)) in enumerate(zip(tool.replay(root, L, R, ctx, copies=B), opk)):
            n_steps += 1
            assert (rec["ts"], rec["k"]) == (ot, ok)
            # the frames the staging slot received are the frames on disk, for every copy
            assert np.array_equal(rec["left"][0], z["lefts"][ok]) and np.array_equal(rec["right"][B - 1], z["rights"][ok])
            # rotation since the last keyframe: oracle's preintegration on the oracle's packet
            dR = np.array(ora.preintegrate_rotation(ostamps, [r[3:6] for r in orows], deltaRij=dR.reshape(9))).reshape(3, 3)
            Rk = body_R_cam.T @ dR @ body_R_cam
            assert np.allclose(Rk, rec["Rk"], atol=1e-15, rtol=0), (i, np.abs(Rk - rec["Rk"]).max())
            exp = fe.process(z["lefts"][ok], z["rights"][ok], ot, rec["Rk"], False)
            assert_step_equal(rec["out"], exp, ("step", i))
            assert_step_equal(ctx.get_output(B - 1), exp, ("step", i, "copy", B - 1))
            if exp["is_keyframe"]:
                dR = np.eye(3)
                n_kf += 1
        assert n_steps == 8
        assert n_kf >= 2      # 20 Hz frames, min_intra_keyframe_time 0.2 s: the first frame and every 4th after it
    finally:
        ctx.close()
