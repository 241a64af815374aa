"""Run configuration: a small typed mirror of the reference harness's YAML (hydra / omegaconf are not available here).

Key names are the ones the reference's reproduction guide shows (``data_path, batch_size, learning_rate, num_iterations,
eval_stage.num_vis, eval_stage.wandb_mode`` -- /root/reference/website/src/pages/[lang]/reprod/index.astro:246-252); the
launcher environment variables are torchrun's (``MASTER_PORT`` etc., index.astro:238-239)."""
from __future__ import annotations

from dataclasses import asdict, dataclass, field, fields, is_dataclass
from typing import Any, Dict

import yaml


@dataclass
class EvalStage:
    num_vis: int = 0
    wandb_mode: str = "offline"


@dataclass
class DataCfg:
    kind: str = "synthetic"          # synthetic | npy_clips | echonet_npz | camus_png
    frames: int = 10
    size: int = 256
    num_classes: int = 4


@dataclass
class ModelCfg:
    heads: int = 1
    value_dim: int = 256
    rule: str = "delta_sequential"
    scan_segments: int = 1           # evaluation of long clips: GDKVMConfig.scan_segments (1 = serial scan, bit-identical under chunking)


@dataclass
class RunConfig:
    data_path: str = ""
    batch_size: int = 8
    learning_rate: float = 1.0e-4
    num_iterations: int = 3000
    eval_stage: EvalStage = field(default_factory=EvalStage)
    data: DataCfg = field(default_factory=DataCfg)
    model: ModelCfg = field(default_factory=ModelCfg)
    run_dir: str = "outputs"
    save_every: int = 1000
    log_every: int = 20
    seed: int = 0
    precision: str = "bf16"

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


def _build(cls, raw: Dict[str, Any]):
    known = {f.name: f for f in fields(cls)}
    unknown = set(raw) - set(known)
    if unknown:
        raise KeyError(f"unknown configuration key(s) for {cls.__name__}: {sorted(unknown)}")
    kw = {}
    for name, val in raw.items():
        ft = known[name].default_factory() if callable(getattr(known[name], "default_factory", None)) and \
            known[name].default_factory is not None and is_dataclass(known[name].default_factory()) else None
        kw[name] = _build(type(ft), val or {}) if ft is not None else val
    return cls(**kw)


def load_config(path: str | None = None, overrides: list[str] | None = None) -> RunConfig:
    """YAML file + ``key=value`` / ``a.b=value`` overrides (the hydra command-line style the reference uses)."""
    raw: Dict[str, Any] = {}
    if path:
        with open(path) as f:
            raw = yaml.safe_load(f) or {}
    for ov in overrides or []:
        key, _, val = ov.partition("=")
        if not _:
            raise ValueError(f"override {ov!r} is not key=value")
        node = raw
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    cfg = _build(RunConfig, raw)
    cfg.learning_rate = float(cfg.learning_rate)
    return cfg
