"""Offline run logging in the shape the reference's guide shows (wandb *offline* mode: ``wandb/offline-run-*`` folders that
are synced later -- /root/reference/website/src/pages/[lang]/reprod/index.astro:274-281).  wandb is not installed in this
image and is not needed: metrics are JSON lines, one record per logged step, plus the resolved config."""
from __future__ import annotations

import json
import os
import time


class OfflineRun:
    def __init__(self, run_dir: str, config: dict, mode: str = "offline", enabled: bool = True):
        self.enabled = enabled and mode != "disabled"
        self.dir = os.path.join(run_dir, "wandb", time.strftime("offline-run-%Y%m%d_%H%M%S"))
        self._f = None
        if self.enabled:
            os.makedirs(self.dir, exist_ok=True)
            with open(os.path.join(self.dir, "config.json"), "w") as f:
                json.dump(config, f, indent=1)
            self._f = open(os.path.join(self.dir, "metrics.jsonl"), "a", buffering=1)

    def log(self, step: int, **metrics):
        if self._f is not None:
            self._f.write(json.dumps({"step": step, "time": time.time(), **metrics}) + "\n")

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None
