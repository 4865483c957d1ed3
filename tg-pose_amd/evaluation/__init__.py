from .metrics import compute_degree_cm_mAP, pair_metrics  # noqa: F401
