"""CPU side of bench.py --scene (scene.prepare_training_json / load_disk_dataset): what is injected into a stock nerf-synthetic file and what is left alone
(this fork loads with NERF_SCALE 1 / offset 0, nerf_loader.h:28, nerf_loader.cu:406-407, 472-474), frame paths, missing frames, test transforms."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]


def _write(tmp_path, meta_extra, n=3, res=(12, 8), drop=None):
    Image = pytest.importorskip("PIL.Image")
    os.makedirs(tmp_path / "train", exist_ok=True)
    os.makedirs(tmp_path / "test", exist_ok=True)
    rs = np.random.RandomState(0)
    frames = []
    for i in range(n):
        if i != drop:
            Image.fromarray(rs.randint(0, 255, (res[1], res[0], 4), dtype=np.uint8), "RGBA").save(str(tmp_path / "train" / ("r_%d.png" % i)))
        m = np.eye(4); m[0, 3] = i
        frames.append({"file_path": "./train/r_%d" % i, "transform_matrix": m.tolist()})
    meta = {"camera_angle_x": 0.69, "frames": frames}
    meta.update(meta_extra)
    train = str(tmp_path / "transforms_train.json")
    json.dump(meta, open(train, "w"))
    Image.fromarray(rs.randint(0, 255, (res[1], res[0], 4), dtype=np.uint8), "RGBA").save(str(tmp_path / "test" / "r_0.png"))
    test = str(tmp_path / "transforms_test.json")
    json.dump({"camera_angle_x": 0.7, "frames": [{"file_path": "./test/r_0", "transform_matrix": np.eye(4).tolist()}]}, open(test, "w"))
    return train, test


def test_stock_file_gets_scale_and_offset_in_a_patched_copy(tmp_path):
    import scene
    train, test = _write(tmp_path, {}, drop=1)
    out, meta, patched = scene.prepare_training_json(train, str(tmp_path / "work"))
    assert patched and out != train and os.path.dirname(out) == str(tmp_path / "work")
    m = json.load(open(out))
    assert m["scale"] == 0.33 and m["offset"] == [0.5, 0.5, 0.5] and m["camera_angle_x"] == 0.69
    assert all(os.path.isabs(f["file_path"]) for f in m["frames"])                       # the copy may live anywhere
    assert json.load(open(train)).keys() == {"camera_angle_x", "frames"}                 # the original is not touched
    ds = scene.load_disk_dataset(train, test, workdir=str(tmp_path / "work2"), decode_train=True)
    assert ds["scale_offset_injected"] and ds["n_train"] == 2                            # the frame whose image is missing is dropped, as the loader drops it (nerf_loader.cu:383)
    assert (ds["w"], ds["h"]) == (12, 8) and ds["camera_angle_x"] == 0.69 and ds["test_camera_angle_x"] == 0.7
    assert len(ds["train_images"]) == 2 and ds["train_images"][0].shape == (8, 12, 4)
    assert [float(p[0, 3]) for p in ds["train_poses"]] == [0.0, 2.0]
    assert len(ds["test_images"]) == 1 and os.path.isfile(ds["test_images"][0]) and ds["test_images"][0].endswith("r_0.png")


@pytest.mark.parametrize("extra", [{"scale": 0.5}, {"offset": [0.1, 0.2, 0.3]}, {"aabb": [[-1, -1, -1], [1, 1, 1]]}])
def test_a_file_that_places_the_scene_itself_is_used_where_it_lies(tmp_path, extra):
    import scene
    train, _ = _write(tmp_path, extra)
    out, meta, patched = scene.prepare_training_json(train, str(tmp_path / "work"))
    assert not patched and out == os.path.abspath(train) and not os.path.exists(str(tmp_path / "work"))
    ds = scene.load_disk_dataset(train, None, workdir=str(tmp_path / "w"))
    assert not ds["scale_offset_injected"] and ds["train_path"] == os.path.abspath(train) and ds["test_images"] == [] and ds["n_train"] == 3
