"""lib/utils/generic.py counterpart: config import by dotted path (:51-56) and the run logger (:7-48)."""
import logging
import time
from pathlib import Path

from ..configs.config_dict import load_config


def import_configs(config_path):
    return load_config(config_path)


def create_logger(cfg, phase="train", no_logger=False, folder_name=""):
    root = Path(cfg.OUTPUT_DIR)
    dataset = (cfg.DATASET.TRAIN_DATASET + "_" + cfg.DATASET.TEST_DATASET).replace(":", "_")
    stamp = time.strftime("%Y-%m-%d-%H-%M-%S")
    out = root / dataset / (f"{stamp}-{folder_name}" if folder_name else stamp)
    if no_logger:
        return None, str(out), None
    out.mkdir(parents=True, exist_ok=True)
    logging.basicConfig(filename=str(out / f"{stamp}_{phase}.log"), format="%(asctime)-15s %(message)s", force=True)
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    logging.getLogger("").addHandler(logging.StreamHandler())
    return logger, str(out), str(out)
