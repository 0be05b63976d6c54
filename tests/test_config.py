"""CPU: the reference's YAML config surface (unet_zoo/config.py:10-49, configs/default_train_config.yaml,
scripts/train.py:141-152) read by unet_zoo_amd.config.Config, plus the two additive keys SURVEY §5 names."""
import copy
import os

import pytest
import torch
import yaml

import unet_zoo_amd
from unet_zoo_amd.config import Config, ConfigError, load_config

# the shipped default_train_config.yaml of the reference, restated as data (keys and values; comments dropped)
DEFAULT_YAML = """
general: {project_name: "UNetZooTraining", working_dir: "./training_runs"}
data: {dataset_dir: "/data/jupyter_folder/pano_unet_bone/bone_mask", num_workers: 4, image_size: 512}
training:
  epochs: 80
  batch_size: 4
  learning_rate: 0.0001
  early_stopping_patience: 20
  lr_scheduler_patience: 8
  lr_scheduler_factor: 0.2
  min_lr: 1e-7
  num_classes: 1
gpu: {use_multi_gpu: false, gpu_ids: [0, 1, 2, 3, 4, 5, 6, 7], single_gpu_id: 0}
models:
  names: [unet, attention_unet, u2net, swin_unet_v2, resunet, nested_unet, missformer]
  params:
    unet: {depth: 5}
    attention_unet: {depth: 4}
    u2net: {}
    swin_unet_v2: {embed_dim: 96, depths: [2, 2, 2, 2], num_heads: [3, 6, 12, 24], window_size: 8, mlp_ratio: 4.0,
                   drop_rate: 0.0, attn_drop_rate: 0.0, drop_path_rate: 0.1}
    resunet: {filters: [64, 128, 256, 512]}
"""


def _cfg(tmp_path, **edits):
    d = yaml.safe_load(DEFAULT_YAML)
    d["general"]["working_dir"] = str(tmp_path)
    for dotted, v in edits.items():
        sec, key = dotted.split("__")
        d[sec][key] = v
    return d


def test_reference_attribute_names_and_defaults(tmp_path):
    c = Config(_cfg(tmp_path))
    assert (c.PROJECT_NAME, c.NUM_WORKERS, c.IMAGE_SIZE, c.EPOCHS, c.BATCH_SIZE) == ("UNetZooTraining", 4, 512, 80, 4)
    assert c.LEARNING_RATE == 1e-4 and c.MIN_LR == 1e-7 and isinstance(c.MIN_LR, float)
    assert (c.EARLY_STOPPING_PATIENCE, c.LR_SCHEDULER_PATIENCE, c.LR_SCHEDULER_FACTOR, c.NUM_CLASSES) == (20, 8, 0.2, 1)
    assert c.USE_MULTI_GPU is False and c.GPU_IDS == list(range(8)) and c.SINGLE_GPU_ID == 0
    assert c.MULTI_GPU_STRATEGY == "DataParallel" and not c.ddp and c.ranks_wanted() == 1
    assert c.RUN_DTYPE == torch.bfloat16
    assert os.path.isdir(c.OVERALL_LOG_DIR) and os.path.isdir(c.TENSORBOARD_BASE_DIR)
    assert c.BASE_RUN_DIR.startswith(str(tmp_path)) and "overall_runs_" in c.BASE_RUN_DIR
    assert c.DEVICE.type == ("cuda" if torch.cuda.device_count() else "cpu")
    assert c.MODELS_TO_TRAIN[:2] == ["unet", "attention_unet"]
    with pytest.raises(KeyError):
        Config({"general": {}})


def test_model_kwargs_are_what_train_py_builds(tmp_path):
    c = Config(_cfg(tmp_path))
    assert c.model_kwargs("unet") == {"depth": 5, "in_channels": 3, "num_classes": 1, "image_size": 512}
    kw = c.model_kwargs("swin_unet_v2")
    assert kw["window_size"] == 8 and kw["image_size"] == 512 and kw["drop_path_rate"] == 0.1
    assert c.model_kwargs("not_in_params") == {"in_channels": 3, "num_classes": 1, "image_size": 512}
    kw["window_size"] = 99                                   # a copy: the config is not mutated
    assert c.model_kwargs("swin_unet_v2")["window_size"] == 8
    m = unet_zoo_amd.create_model("unet", **c.model_kwargs("unet"))
    assert sum(p.numel() for p in m.parameters()) == 31_043_521
    m = unet_zoo_amd.create_model("resunet", **c.model_kwargs("resunet"))
    assert sum(p.numel() for p in m.parameters()) == 8_048_705


def test_additive_keys_dtype_and_ddp_rccl(tmp_path, monkeypatch):
    c = Config(_cfg(tmp_path, training__dtype="fp32"))
    assert c.RUN_DTYPE == torch.float32
    with pytest.raises(ConfigError):
        Config(_cfg(tmp_path, training__dtype="fp8"))
    c = Config(_cfg(tmp_path, gpu__use_multi_gpu=True, gpu__multi_gpu_strategy="ddp_rccl", gpu__gpu_ids=[0, 1, 2, 3]))
    assert c.ddp and c.ranks_wanted() == 4
    with pytest.raises(ConfigError):                         # single-process DataParallel over several GPUs
        Config(_cfg(tmp_path, gpu__use_multi_gpu=True))
    with pytest.raises(ConfigError):
        Config(_cfg(tmp_path, gpu__multi_gpu_strategy="fsdp"))
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    c = Config(_cfg(tmp_path / "r1", gpu__use_multi_gpu=True, gpu__multi_gpu_strategy="ddp_rccl", gpu__gpu_ids=[0, 1]))
    assert (c.RANK, c.LOCAL_RANK, c.WORLD_SIZE) == (1, 1, 2)
    assert not os.path.exists(c.OVERALL_LOG_DIR)             # rank 0 creates the run directories
    # gpu.gpu_ids places the ranks (rank r on gpu_ids[r], as the reference places its replicas); every rank of one
    # launch derives the same run directory from the launcher's UZ_RUN_TIMESTAMP
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("UZ_RUN_TIMESTAMP", "20260101-000000")
    c = Config(_cfg(tmp_path / "r1b", gpu__use_multi_gpu=True, gpu__multi_gpu_strategy="ddp_rccl", gpu__gpu_ids=[2, 3]))
    assert c.DEVICE == torch.device("cuda", 3)
    assert c.RUN_TIMESTAMP == "20260101-000000" and c.BASE_RUN_DIR.endswith("overall_runs_20260101-000000")
    with pytest.raises(ValueError):
        Config(_cfg(tmp_path / "r1c", gpu__use_multi_gpu=True, gpu__multi_gpu_strategy="ddp_rccl", gpu__gpu_ids=[0, 1, 2]))


def test_load_config_from_file(tmp_path):
    p = tmp_path / "train.yaml"
    d = _cfg(tmp_path)
    p.write_text(yaml.safe_dump(d))
    c = load_config(str(p), run_timestamp="20260101-000000", make_dirs=False)
    assert c.RUN_TIMESTAMP == "20260101-000000" and not os.path.exists(c.BASE_RUN_DIR)
    m = c.create("unet") if c.DEVICE.type == "cuda" else unet_zoo_amd.create_model("unet", **c.model_kwargs("unet"))
    assert m.run_dtype == torch.bfloat16
