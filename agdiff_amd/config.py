"""Attribute-style config object (stands where the reference uses easydict.EasyDict,
scripts/test.py:81-82) and the two model blocks of configs/{qm9,drugs}_default.yml:1-15."""


class Config(dict):
    """dict with attribute access; nested dicts become Config on construction."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, Config):
                self[k] = Config(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


_MODEL_COMMON = dict(
    type="diffusion", network="dualenc", hidden_dim=128, num_convs=6, num_convs_local=4,
    cutoff=10.0, mlp_act="relu", beta_schedule="sigmoid", beta_start=1.0e-7, beta_end=2.0e-3,
    num_diffusion_timesteps=5000, edge_order=3, edge_encoder="mlp",
)


def qm9_model_config(**over):
    """configs/qm9_default.yml:1-15 (smooth_conv: false)."""
    return Config(dict(_MODEL_COMMON, smooth_conv=False, **over))


def drugs_model_config(**over):
    """configs/drugs_default.yml:1-15 (smooth_conv: true)."""
    return Config(dict(_MODEL_COMMON, smooth_conv=True, **over))
