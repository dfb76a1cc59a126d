from icafusion_amd.models.experimental import attempt_load, Ensemble  # noqa: F401
