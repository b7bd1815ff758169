"""The C ABI without Python or torch on the calling side: examples/c_host/sttran_c_host.c (plain C11, gcc) loads the
weights, owns the device buffers, calls sttran_forward on its own stream -- and must reproduce the Python shim's
outputs bit for bit (same library, same kernels, same tile plans)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from nl_vsgg_amd.lib import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_host", "sttran_c_host.c")
CSRC = os.path.join(ROOT, "nl-vsgg_amd", "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp):
    exe = os.path.join(tmp, "sttran_c_host")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROCM, "include"), SRC, "-L", CSRC, "-lsttran_hip", "-L", os.path.join(ROCM, "lib"),
           "-lamdhip64", f"-Wl,-rpath,{CSRC}", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_c_host_compiles_as_plain_c(tmp_path):
    """CPU: the header is valid C (not only C++) and the example links against the library."""
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(CSRC, "libsttran_hip.so")):
        pytest.skip("gcc or the built library is missing")
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_host_matches_python_shim(tmp_path):
    import torch
    from nl_vsgg_amd.lib.sttran import STTran
    exe = _build(str(tmp_path))
    sd = syn.make_sttran_state_dict(7)
    counts = [3, 0, 5, 2, 4]
    e = syn.make_entry(808, counts)
    # ---- bundle the weights and the entry in the example's file formats
    wpath, epath, opath = (str(tmp_path / n) for n in ("weights.bin", "entry.bin", "out.bin"))
    with open(wpath, "wb") as f:
        keys = [k for k, v in sd.items() if np.asarray(v).dtype == np.float32]
        f.write(struct.pack("<i", len(keys)))
        for k in keys:
            v = np.ascontiguousarray(sd[k], dtype=np.float32)
            kb = k.encode()
            f.write(struct.pack("<i", len(kb))); f.write(kb)
            f.write(struct.pack("<i", v.ndim)); f.write(np.asarray(v.shape, dtype=np.int64).tobytes())
            f.write(v.tobytes())
    B, P, T = e["features"].shape[0], e["pair_idx"].shape[0], len(counts)
    with open(epath, "wb") as f:
        f.write(struct.pack("<iqqi", 0, B, P, T))
        f.write(np.asarray(counts, dtype=np.int32).tobytes())
        for k, dt in (("features", np.float32), ("pair_idx", np.int64), ("labels", np.int64), ("union_feat", np.float32),
                      ("spatial_masks", np.float32), ("im_idx", np.float32)):
            f.write(np.ascontiguousarray(e[k], dtype=dt).tobytes())
    o2path = str(tmp_path / "out2.bin")
    r = subprocess.run([exe, wpath, epath, opath, o2path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # the example also ran the clip on two lanes of the handle and compared, in C, with its classic call
    assert "lanes: 2 lanes, results identical to the classic forward" in r.stdout, r.stdout
    got = np.fromfile(opath, dtype=np.float32)
    assert got.size == P * 26
    att, spa, con = got[:P * 3].reshape(P, 3), got[P * 3:P * 9].reshape(P, 6), got[P * 9:].reshape(P, 17)
    # ---- the same clip through the Python shim
    m = STTran(mode="predcls", attention_class_num=3, spatial_class_num=6, contact_class_num=17,
               obj_classes=["__background__"] + [f"c{i}" for i in range(36)], enc_layer_num=1, dec_layer_num=3,
               transformer_mode="wk", is_wks=True, feat_dim=2048).to("cuda:0")
    m.eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    pred = m({k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()})
    torch.cuda.synchronize()
    np.testing.assert_array_equal(att, pred["attention_distribution"].cpu().numpy())
    np.testing.assert_array_equal(spa, pred["spatial_distribution"].cpu().numpy())
    np.testing.assert_array_equal(con, pred["contacting_distribution"].cpu().numpy())
    # ---- the two-clip call by pointer tables == the shim's pack_clips(copy=False) of the same two clips
    from nl_vsgg_amd.lib.sttran import pack_clips
    ce = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) and k != "frame_counts" else v) for k, v in e.items()}
    two = m(pack_clips([ce, dict(ce)], copy=False))
    torch.cuda.synchronize()
    got2 = np.fromfile(o2path, dtype=np.float32)
    assert got2.size == 2 * P * 26
    np.testing.assert_array_equal(got2[:2 * P * 3].reshape(2 * P, 3), two["attention_distribution"].cpu().numpy())
    np.testing.assert_array_equal(got2[2 * P * 3:2 * P * 9].reshape(2 * P, 6), two["spatial_distribution"].cpu().numpy())
    np.testing.assert_array_equal(got2[2 * P * 9:].reshape(2 * P, 17), two["contacting_distribution"].cpu().numpy())
