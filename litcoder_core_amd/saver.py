"""On-disk result format of the reference's ``ModelSaver`` (``encoding/utils.py:288-414``), the files downstream
LITcoder tooling reads.  Host-only Python, same names and arguments:

    <base_dir>/run_<YYYYmmdd_HHMMSS>_<md5(json.dumps(hyperparams, sort_keys=True))[:8]>/
        hyperparams.json   json.dump(hyperparams, indent=2)              (utils.py:318-319)
        metrics.pkl        pickle.dump(metrics)                          (utils.py:349-350)
        weights.npy        np.save(weights), only with save_weights=True (utils.py:345-346)

One deviation, on purpose: the reference's ``load_encoding_model`` reads ``best_alphas.npy`` (utils.py:370), a file its
own ``save_encoding_model`` never writes, so it cannot load what it saved.  Here the alphas come from that file when
it exists and from ``metrics["best_alphas"]`` otherwise.
"""
import hashlib
import json
import pickle
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, List, Tuple, Union

import numpy as np


class ModelSaver:
    def __init__(self, base_dir: str = "results"):
        self.base_dir = Path(base_dir)
        self.base_dir.mkdir(parents=True, exist_ok=True)

    @staticmethod
    def run_hash(hyperparams: Dict[str, Any]) -> str:
        """utils.py:308-309: first 8 hex digits of the md5 of the key-sorted JSON text."""
        return hashlib.md5(json.dumps(hyperparams, sort_keys=True).encode()).hexdigest()[:8]

    def _create_run_dir(self, hyperparams: Dict[str, Any]) -> Path:
        timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        run_dir = self.base_dir / f"run_{timestamp}_{self.run_hash(hyperparams)}"
        run_dir.mkdir(parents=True, exist_ok=True)
        with open(run_dir / "hyperparams.json", "w") as f:
            json.dump(hyperparams, f, indent=2)
        return run_dir

    def save_encoding_model(self, weights: np.ndarray, best_alphas: np.ndarray, hyperparams: Dict[str, Any],
                            metrics: Dict[str, Any], save_weights: bool = False) -> Path:
        run_dir = self._create_run_dir(hyperparams)
        if save_weights:
            np.save(run_dir / "weights.npy", weights)
        with open(run_dir / "metrics.pkl", "wb") as f:
            pickle.dump(metrics, f)
        return run_dir

    def load_encoding_model(self, run_dir: Union[str, Path]) -> Tuple[np.ndarray, np.ndarray, Dict[str, Any],
                                                                     Dict[str, Any]]:
        run_dir = Path(run_dir)
        weights = np.load(run_dir / "weights.npy")
        with open(run_dir / "hyperparams.json", "r") as f:
            hyperparams = json.load(f)
        with open(run_dir / "metrics.pkl", "rb") as f:
            metrics = pickle.load(f)
        alphas_file = run_dir / "best_alphas.npy"
        best_alphas = np.load(alphas_file) if alphas_file.exists() else np.asarray(metrics["best_alphas"])
        return weights, best_alphas, hyperparams, metrics

    def list_runs(self) -> List[Dict[str, Any]]:
        runs = []
        for run_dir in self.base_dir.glob("run_*"):
            if not run_dir.is_dir():
                continue
            try:
                with open(run_dir / "hyperparams.json", "r") as f:
                    hyperparams = json.load(f)
                with open(run_dir / "metrics.pkl", "rb") as f:
                    metrics = pickle.load(f)
                runs.append({"run_dir": str(run_dir), "timestamp": run_dir.name.split("_")[1],
                             "hyperparams": hyperparams, "metrics": metrics})
            except Exception as e:  # the reference prints and skips unreadable runs (utils.py:405-407)
                print(f"Error loading run {run_dir}: {e}")
                continue
        runs.sort(key=lambda x: x["timestamp"], reverse=True)
        return runs
