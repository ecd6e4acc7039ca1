"""Configuration values of the hot path (mirrors core/configs.py).

The keys that shape the forward pass, the losses and the optimiser schedule are kept with the reference's
values (core/configs.py:35-86 basic_config, :88-102 detection_config, :104-144 global_config); dataset,
augmentation and logging keys are out of scope.  Missing keys read as None, like the reference's dotdict
(core/configs.py:22-26).  `fps_contract` is an extension (None = default FPS kernels; 0 / 1 = forced
distance rounding, see ops.farthest_point_sample).
"""


class dotdict(dict):
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


class ConfigFactory(object):
    def __init__(self, name="basic_config"):
        self.config_name = name

    def basic_config(self):
        return {
            "extract_global": False,
            "detection": False,
            "local_backbone": "backbone_local_dilate",
            "add_batch_norm": True,
            "init_feat_dim": 32,
            "featdim": 128,
            "knn_num": 8,
            "dilate": 8,
            "num_points": 8192,
            "batch_size": 10,
            "num_pos": 1,
            "num_neg": 0,
            "other_neg": False,
            "sampled_kpnum": 512,
            # training switches / schedule (core/configs.py:38-54)
            "training_local": True, "freezedetection": False, "freezebackbone": False, "freezeglobal": False,
            "start_lr": 5e-4, "decay_step": 5 * 2000, "decay_rate": 0.5,
            "add_weight_decay": True, "train_weight_decay": 1e-5,
            # losses (core/configs.py:72-82)
            "add_local_loss": True, "add_det_loss": False, "add_global_loss": False,
            "margin": 1.0, "neg_weight": 5.0, "local_loss": "desc_local_loss", "pos_r": 0.5, "search_r": 20.0,
            "local_loss_weight": 1.0,
            "fps_contract": None,
            # BatchNorm epsilons of the (un-vendored) third-party layers; see DESIGN.md
            "tp_bn_eps": 1e-5,    # tensorpack BatchNorm default
            "slim_bn_eps": 1e-3,  # tf.contrib.slim / tf.contrib.layers batch_norm default
        }

    def detection_config(self):
        cfg = self.basic_config()
        cfg.update({"detection": True, "detection_block": "detection_block", "add_det_loss": True,
                    "detection_loss": "local_detection_loss_nn", "ar_th": 0.4, "det_k": 16, "ar_nn_k": 5,
                    "det_loss_weight": 0.2})
        return cfg

    def global_config(self):
        cfg = self.basic_config()
        cfg.update({
            "extract_global": True,
            "detection": False,
            "training_local": False, "freezebackbone": True, "freezedetection": True,
            "start_lr": 5e-4, "decay_step": 20000, "decay_rate": 0.9,
            "sampled_kpnum": -1,
            "add_local_loss": False, "add_det_loss": False, "add_global_loss": True, "global_loss_weight": 1,
            "global_backbone": "global_before_assemble",
            "global_assemble": "global_netvald_block",
            "concat_xyz": False,
            "global_subsample": -1,
            "gl_dilate": 8,
            "gl_dims": [256],
            "batch_size": 2,
            "num_pos": 2,
            "num_neg": 8,
            "other_neg": True,
            "global_loss": "lazy_quadruplet_loss",
            "global_triplet_margin": 0.5,
            "global_quadruplet_margin": 0.2,
        })
        return cfg

    def getconfig(self):
        cfg = self.basic_config()
        cfg.update(getattr(self, self.config_name)())
        return dotdict(cfg)
