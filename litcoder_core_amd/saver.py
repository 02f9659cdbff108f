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

    # ---- readers (own structure; the WRITTEN layout above is what SURVEY 8f-4 asks for, these are conveniences with the
    # reference's names and return shapes: utils.py:352-414)
    @staticmethod
    def _read_run(run_dir: Path, want_weights: bool):
        """(hyperparams, metrics, weights or None) of one run directory."""
        hyper = json.loads((run_dir / "hyperparams.json").read_text())
        metrics = pickle.loads((run_dir / "metrics.pkl").read_bytes())
        weights = np.load(run_dir / "weights.npy") if want_weights else None
        return hyper, metrics, weights

    def load_encoding_model(self, run_dir: Union[str, Path]) -> Tuple[np.ndarray, np.ndarray, Dict[str, Any],
                                                                     Dict[str, Any]]:
        """(weights, best_alphas, hyperparams, metrics) of a run saved with ``save_weights=True``."""
        run_dir = Path(run_dir)
        hyper, metrics, weights = self._read_run(run_dir, want_weights=True)
        stored = run_dir / "best_alphas.npy"           # the reference reads this file but never writes it (see above)
        alphas = np.load(stored) if stored.exists() else np.asarray(metrics["best_alphas"])
        return weights, alphas, hyper, metrics

    def list_runs(self) -> List[Dict[str, Any]]:
        """One record per readable ``run_*`` directory, newest first; an unreadable run is reported and left out."""
        records = []
        for run_dir in sorted((d for d in self.base_dir.glob("run_*") if d.is_dir()),
                              key=lambda d: d.name.split("_")[1], reverse=True):
            try:
                hyper, metrics, _ = self._read_run(run_dir, want_weights=False)
            except Exception as exc:  # noqa: BLE001 -- utils.py:405-407 prints and goes on
                print(f"Error loading run {run_dir}: {exc}")
                continue
            records.append(dict(run_dir=str(run_dir), timestamp=run_dir.name.split("_")[1], hyperparams=hyper,
                                metrics=metrics))
        return records
