"""The reference's YAML config surface (SURVEY.md §5 "Config / flags"; unet_zoo/config.py:10-49,
configs/default_train_config.yaml:1-90, scripts/train.py:58-64, 141-152).

``Config(yaml_dict)`` exposes the attribute names the reference's scripts read (``BATCH_SIZE``, ``LEARNING_RATE``,
``GPU_IDS``, ``DEVICE`` ...), so ``scripts/train.py`` keeps working when its ``from unet_zoo.config import Config`` is
pointed here, plus what the HIP engine adds:

* ``training.dtype`` (``bf16`` | ``fp32``, default ``bf16``) -> ``RUN_DTYPE``; applied to models by :meth:`create`;
* ``gpu.multi_gpu_strategy: ddp_rccl`` (the key already exists in the reference, config.py:30, with the single value
  ``DataParallel``): one process per GPU, RCCL gradient all-reduce (``unet_zoo_amd.launch`` / ``GraphedStep``).
  ``DataParallel`` is rejected on a multi-GPU request: that scheme is what this engine replaces;
* ``models.names`` / ``models.params`` -> ``MODELS_TO_TRAIN`` / ``MODEL_PARAMS`` and :meth:`model_kwargs`, the
  dictionary scripts/train.py:146-150 builds for ``create_model``.

Host-side only: no CLI, logger, scheduler or dataset classes live here.
"""
from __future__ import annotations

import copy
import datetime
import os
from typing import Any, Dict, List, Optional

import torch

DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}
STRATEGIES = ("dataparallel", "ddp_rccl")


class ConfigError(ValueError):
    pass


def _section(d: dict, name: str) -> dict:
    if name not in d or not isinstance(d[name], dict):
        raise KeyError(name)            # the reference indexes the sections directly: a missing one is a KeyError
    return d[name]


class Config:
    def __init__(self, overall_config_dict: Dict[str, Any], make_dirs: bool = True):
        cfg = overall_config_dict
        general, data = _section(cfg, "general"), _section(cfg, "data")
        training, gpu = _section(cfg, "training"), _section(cfg, "gpu")

        self.PROJECT_NAME = general["project_name"]
        self.WORKING_DIR = general["working_dir"]

        self.DATASET_DIR = data["dataset_dir"]
        self.NUM_WORKERS = data["num_workers"]
        self.IMAGE_SIZE = data.get("image_size", 512)

        for attr, key in (("EPOCHS", "epochs"), ("BATCH_SIZE", "batch_size"), ("LEARNING_RATE", "learning_rate"),
                          ("EARLY_STOPPING_PATIENCE", "early_stopping_patience"),
                          ("LR_SCHEDULER_PATIENCE", "lr_scheduler_patience"),
                          ("LR_SCHEDULER_FACTOR", "lr_scheduler_factor"), ("MIN_LR", "min_lr"),
                          ("NUM_CLASSES", "num_classes")):
            setattr(self, attr, training[key])
        # yaml.safe_load reads "1e-7" (no dot) as a string; the reference passes it on as is, numbers are meant
        for attr in ("LEARNING_RATE", "MIN_LR", "LR_SCHEDULER_FACTOR"):
            v = getattr(self, attr)
            if isinstance(v, str):
                setattr(self, attr, float(v))

        dt = str(training.get("dtype", "bf16")).lower()
        if dt not in DTYPES:
            raise ConfigError(f"training.dtype must be one of {sorted(DTYPES)}, got {dt!r}")
        self.RUN_DTYPE = DTYPES[dt]

        self.USE_MULTI_GPU = gpu["use_multi_gpu"]
        self.GPU_IDS = list(gpu["gpu_ids"])
        self.SINGLE_GPU_ID = gpu["single_gpu_id"]
        self.MULTI_GPU_STRATEGY = gpu.get("multi_gpu_strategy", "DataParallel")
        if str(self.MULTI_GPU_STRATEGY).lower() not in STRATEGIES:
            raise ConfigError(f"gpu.multi_gpu_strategy must be 'ddp_rccl' (or the reference's 'DataParallel' with "
                              f"use_multi_gpu: false), got {self.MULTI_GPU_STRATEGY!r}")
        if self.USE_MULTI_GPU and len(self.GPU_IDS) > 1 and str(self.MULTI_GPU_STRATEGY).lower() != "ddp_rccl":
            raise ConfigError("use_multi_gpu with the reference's single-process nn.DataParallel is not offered by the "
                              "HIP engine: set gpu.multi_gpu_strategy: ddp_rccl and launch one process per GPU "
                              "(python -m unet_zoo_amd.launch --gpus N ...)")

        self.WORLD_SIZE = int(os.environ.get("WORLD_SIZE", "1"))
        self.RANK = int(os.environ.get("RANK", "0"))
        self.LOCAL_RANK = int(os.environ.get("LOCAL_RANK", "0"))
        self.DEVICE = self._pick_device()

        models = cfg.get("models", {}) or {}
        self.MODELS_TO_TRAIN: List[str] = list(models.get("names", []) or [])
        self.MODEL_PARAMS: Dict[str, Dict[str, Any]] = copy.deepcopy(models.get("params", {}) or {})

        # every rank of one launch must agree on the run directory: the launcher exports one UZ_RUN_TIMESTAMP
        # (launch.rank_env); ranks started across a second boundary would otherwise derive different names
        self.RUN_TIMESTAMP = (cfg.get("run_timestamp") or os.environ.get("UZ_RUN_TIMESTAMP")
                              or datetime.datetime.now().strftime("%Y%m%d-%H%M%S_fallback"))
        self.BASE_RUN_DIR = os.path.join(self.WORKING_DIR, f"overall_runs_{self.RUN_TIMESTAMP}")
        self.OVERALL_LOG_DIR = os.path.join(self.BASE_RUN_DIR, "overall_logs")
        self.TENSORBOARD_BASE_DIR = os.path.join(self.BASE_RUN_DIR, "tensorboard_logs")
        if make_dirs and self.RANK == 0:
            os.makedirs(self.OVERALL_LOG_DIR, exist_ok=True)
            os.makedirs(self.TENSORBOARD_BASE_DIR, exist_ok=True)

    # ------------------------------------------------------------------ devices
    @property
    def ddp(self) -> bool:
        """one process per GPU with RCCL gradient all-reduce"""
        return bool(self.USE_MULTI_GPU) and str(self.MULTI_GPU_STRATEGY).lower() == "ddp_rccl"

    def ranks_wanted(self) -> int:
        """processes `unet_zoo_amd.launch` should start for this config"""
        return len(self.GPU_IDS) if self.ddp and self.GPU_IDS else 1

    def _pick_device(self) -> torch.device:
        n = torch.cuda.device_count()          # counting does not initialise the runtime
        if n == 0:
            return torch.device("cpu")         # models raise on a CPU input: there is no fallback path
        if self.ddp and self.WORLD_SIZE > 1:
            # one rank per entry of gpu.gpu_ids, as the reference places its replicas (GPU_IDS, multi_gpu.py:20-31):
            # rank r runs on gpu_ids[r]; a list of another length is a configuration error, not something to guess at
            if self.GPU_IDS:
                if len(self.GPU_IDS) != self.WORLD_SIZE:
                    raise ValueError(f"gpu.gpu_ids has {len(self.GPU_IDS)} entries for {self.WORLD_SIZE} ranks")
                return torch.device("cuda", self.GPU_IDS[self.LOCAL_RANK])
            return torch.device("cuda", self.LOCAL_RANK)
        if self.USE_MULTI_GPU and self.GPU_IDS:
            return torch.device("cuda", self.GPU_IDS[0])
        if self.SINGLE_GPU_ID is not None and n > self.SINGLE_GPU_ID:
            return torch.device("cuda", self.SINGLE_GPU_ID)
        return torch.device("cuda", 0)

    def get_device_info(self) -> str:
        if self.DEVICE.type == "cuda":
            return f"CUDA ({torch.cuda.get_device_name(self.DEVICE)})"
        return "CPU"

    # ------------------------------------------------------------------ models
    def model_kwargs(self, name: str) -> Dict[str, Any]:
        """models.params[name] with the three defaults scripts/train.py:148-150 injects"""
        kw = copy.deepcopy(self.MODEL_PARAMS.get(name, {}) or {})
        kw.setdefault("in_channels", 3)
        kw.setdefault("num_classes", self.NUM_CLASSES)
        kw.setdefault("image_size", self.IMAGE_SIZE)
        return kw

    def create(self, name: str):
        """create_model(name, **model_kwargs(name)) in the configured run dtype, on the configured device"""
        from .models import create_model
        model = create_model(name, **self.model_kwargs(name))
        if hasattr(model, "run_dtype"):
            model.run_dtype = self.RUN_DTYPE
        return model.to(self.DEVICE)


def load_config(path: str, run_timestamp: Optional[str] = None, make_dirs: bool = True) -> Config:
    """yaml.safe_load + Config, as scripts/train.py:58-64 does"""
    import yaml
    with open(path, "r") as f:
        d = yaml.safe_load(f)
    if run_timestamp is not None:
        d["run_timestamp"] = run_timestamp
    return Config(d, make_dirs=make_dirs)
