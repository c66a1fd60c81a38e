"""Configuration values of the shipped experiments, restated as plain attribute dictionaries.

The reference builds yacs CfgNodes from configs/**.py (utils/config_utils.py:6-19); the build ships
no yacs, so the values the hot path reads (`cfg.model.*`, `cfg.loss.*`, `cfg.data.*`,
`cfg.optimizer.*`, `cfg.exp.*`) are restated here, each citing its reference file.
"""
from __future__ import annotations


class Config(dict):
    """Attribute-style nested dict exposing the CfgNode calls the models use (`get`, `clone`)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        self[key] = value

    def clone(self):
        return Config({k: (v.clone() if isinstance(v, Config) else v) for k, v in self.items()})


def geometric_loss():
    """configs/_base_/models/loss/geometric_loss.py:17-27."""
    return Config(noise_dim=0, trans_loss_w=1.0, rot_pt_cd_loss_w=10.0, transform_pt_cd_loss_w=10.0,
                  use_rot_loss=True, rot_loss_w=0.2, use_rot_pt_l2_loss=True, rot_pt_l2_loss_w=1.0)


def semantic_loss():
    """configs/_base_/models/loss/semantic_loss.py:12-23."""
    return Config(noise_dim=32, sample_iter=5, trans_loss_w=1.0, rot_pt_cd_loss_w=10.0,
                  transform_pt_cd_loss_w=10.0, use_rot_loss=False, use_rot_pt_l2_loss=False)


def breaking_bad_everyday():
    """configs/_base_/datasets/breaking_bad/everyday.py:5-16."""
    return Config(dataset="geometry", data_keys=("part_ids",), num_pc_points=1000, min_num_part=2,
                  max_num_part=20)


def adam_cosine():
    """configs/_base_/schedules/adam_cosine.py:5-11."""
    return Config(lr=1e-3, weight_decay=0.0, warmup_ratio=0.0, clip_grad=None, lr_scheduler="cosine",
                  lr_decay_factor=100.0)


def pn_transformer_model():
    """configs/_base_/models/pn_transformer/pn_transformer.py:5-15."""
    return Config(name="pn_transformer", rot_type="quat", pc_feat_dim=256, encoder="pointnet",
                  transformer_feat_dim=1024, transformer_heads=8, transformer_layers=4,
                  transformer_pre_ln=True)


def pn_transformer_everyday():
    """configs/pn_transformer/pn_transformer/pn_transformer-32x1-cosine_400e-everyday.py."""
    opt = adam_cosine()
    opt.warmup_ratio = 0.05
    return Config(exp=Config(batch_size=32, num_epochs=400, num_workers=8, gpus=[0]),
                  data=breaking_bad_everyday(), optimizer=opt, model=pn_transformer_model(),
                  loss=geometric_loss())
