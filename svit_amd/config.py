"""cfg tree for the SViT hot path.

The build preserves the reference's cfg/yaml *schema* (slowfast/config/defaults.py,
configs/ssv2.yaml): `get_cfg()` returns a yacs-style node holding the defaults of every key the
hot path reads, `merge_from_file` / `merge_from_list` reproduce the two yacs coercions the
reference relies on: string values are `literal_eval`-ed
(`PATCH_KERNEL: (3, 7, 7)` is a YAML string) and ints/strings are coerced to float where the
default is a float (`BASE_LR: 2e-4` is a YAML 1.1 string).  Keys this build does not know are
kept as they are, so the reference's full configs/ssv2.yaml merges without error.

`ssv2_cfg()` reproduces the SViT settings of the reference's configs/ssv2.yaml programmatically
(values per SURVEY.md sections 5/8 and Appendix A/C); a reference checkout's own yaml can be
merged on top with `cfg.merge_from_file(path)`.
"""
import ast
import copy


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _coerce(new, old):
        if isinstance(new, str):
            try:
                new = ast.literal_eval(new)
            except (ValueError, SyntaxError):
                pass
        if old is None or new is None or type(old) is type(new):
            return new
        if isinstance(old, float) and isinstance(new, (int, str)):
            return float(new)
        if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
            return type(old)(new)
        return new

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], dict):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v, self.get(k))

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("override list must be KEY VAL pairs")
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("non-existent config key: %s" % key)
            node[parts[-1]] = self._coerce(val, node[parts[-1]])

    def dump(self):
        import yaml

        def plain(n):
            return {k: plain(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v)
                    for k, v in n.items()}
        return yaml.safe_dump(plain(self))


def get_cfg():
    """Defaults of the keys on the SViT path (values of slowfast/config/defaults.py)."""
    c = CfgNode()
    c.NUM_GPUS = 1
    c.NUM_SHARDS = 1
    c.SHARD_ID = 0
    c.RNG_SEED = 1
    c.OUTPUT_DIR = "./tmp"
    c.DIST_BACKEND = "nccl"
    c.DDP_FIND_UNUSED_PARAMETERS = False
    c.LOG_PERIOD = 10
    # CONSISTENCY is this build's switch for the paper's frame-clip consistency loss: "" keeps the
    # as-released behaviour (the frames pass runs, its output is unused),
    # "l1" / "l2" weight |obj_desc(video) - obj_desc(frames)| with LAMBDA_CON (losses.py:127-136)
    c.SVIT = CfgNode({"O": 4, "LAMBDA_NODES": 1.0, "LAMBDA_EDGES": 1.0, "LAMBDA_CON": 1.0,
                      "CONSISTENCY": ""})
    c.DATA = CfgNode({"NUM_FRAMES": 8, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256,
                      "INPUT_CHANNEL_NUM": [3, 3], "MEAN": [0.45, 0.45, 0.45],
                      "STD": [0.225, 0.225, 0.225], "SAMPLING_RATE": 8})
    c.MODEL = CfgNode({"ARCH": "slowfast", "MODEL_NAME": "SlowFast", "NUM_CLASSES": 400,
                       "LOSS_FUNC": "cross_entropy", "DROPOUT_RATE": 0.5, "HEAD_ACT": "softmax",
                       "LOAD_IN_PRETRAIN": ""})
    c.MVIT = CfgNode({
        "MODE": "conv", "POOL_FIRST": False, "CLS_EMBED_ON": True, "PATCH_KERNEL": [3, 7, 7],
        "PATCH_STRIDE": [2, 4, 4], "PATCH_PADDING": [2, 4, 4], "PATCH_2D": False,
        "EMBED_DIM": 96, "NUM_HEADS": 1, "MLP_RATIO": 4.0, "QKV_BIAS": True,
        "DROPPATH_RATE": 0.1, "DEPTH": 16, "NORM": "layernorm", "DIM_MUL": [], "HEAD_MUL": [],
        "POOL_KV_STRIDE": [], "POOL_KV_STRIDE_ADAPTIVE": None, "POOL_Q_STRIDE": [],
        "POOL_KVQ_KERNEL": None, "ZERO_DECAY_POS_CLS": True, "NORM_STEM": False,
        "SEP_POS_EMBED": False, "DROPOUT_RATE": 0.0, "USE_ABS_POS": True,
        "REL_POS_SPATIAL": False, "REL_POS_TEMPORAL": False, "REL_POS_ZERO_INIT": False,
        "RESIDUAL_POOLING": False, "DIM_MUL_IN_ATT": False, "SEPARATE_QKV": False})
    c.DETECTION = CfgNode({"ENABLE": False})
    c.TRAIN = CfgNode({"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 64,
                       "MIXED_PRECISION": False, "FORWARD_VIDEO_FRAMES": True})
    c.TEST = CfgNode({"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 8,
                      "NUM_ENSEMBLE_VIEWS": 10, "NUM_SPATIAL_CROPS": 3})
    c.IMAGE_TRAIN = CfgNode({"GPU_IDS": [], "BATCH_SIZE": 64, "DATASETS": []})
    c.BN = CfgNode({"WEIGHT_DECAY": 0.0})
    c.SOLVER = CfgNode({
        "BASE_LR": 0.1, "LR_POLICY": "cosine", "COSINE_END_LR": 0.0, "MAX_EPOCH": 300,
        "MOMENTUM": 0.9, "DAMPENING": 0.0, "NESTEROV": True, "WEIGHT_DECAY": 1e-4,
        "WARMUP_EPOCHS": 0.0, "WARMUP_START_LR": 0.01, "OPTIMIZING_METHOD": "sgd",
        "BASE_LR_SCALE_NUM_SHARDS": False, "COSINE_AFTER_WARMUP": False,
        "ZERO_WD_1D_PARAM": False, "CLIP_GRAD_VAL": None, "CLIP_GRAD_L2NORM": None})
    return c


def ssv2_cfg(num_frames=16, crop=224, num_gpus=1):
    """The SViT recipe of the reference's configs/ssv2.yaml (model + solver keys)."""
    c = get_cfg()
    c.NUM_GPUS = num_gpus
    c.RNG_SEED = 0
    c.DATA.NUM_FRAMES = num_frames
    c.DATA.TRAIN_CROP_SIZE = crop
    c.DATA.TEST_CROP_SIZE = crop
    c.DATA.INPUT_CHANNEL_NUM = [3]
    c.DATA.SAMPLING_RATE = 2
    c.MODEL.ARCH = "mvit"
    c.MODEL.MODEL_NAME = "SViT"
    c.MODEL.NUM_CLASSES = 174
    c.MODEL.LOSS_FUNC = "video_image_loss"
    c.MODEL.DROPOUT_RATE = 0.5
    mv = c.MVIT
    mv.PATCH_KERNEL, mv.PATCH_STRIDE, mv.PATCH_PADDING = [3, 7, 7], [2, 4, 4], [1, 3, 3]
    mv.DEPTH, mv.EMBED_DIM, mv.NUM_HEADS, mv.DROPPATH_RATE = 16, 96, 1, 0.4
    mv.DIM_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
    mv.HEAD_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
    mv.DIM_MUL_IN_ATT = True
    mv.POOL_KVQ_KERNEL = [3, 3, 3]
    mv.POOL_KV_STRIDE_ADAPTIVE = [1, 8, 8]
    mv.POOL_Q_STRIDE = [[i, 1, 2, 2] if i in (1, 3, 14) else [i, 1, 1, 1] for i in range(16)]
    mv.REL_POS_SPATIAL = mv.REL_POS_TEMPORAL = mv.RESIDUAL_POOLING = True
    mv.USE_ABS_POS = False
    mv.ZERO_DECAY_POS_CLS = False
    c.SVIT.LAMBDA_NODES, c.SVIT.LAMBDA_EDGES, c.SVIT.LAMBDA_CON = 3.7, 0.3, 1.5
    c.TRAIN.DATASET = c.TEST.DATASET = "ssv2"
    c.TRAIN.BATCH_SIZE, c.TRAIN.MIXED_PRECISION = 63, True
    c.TEST.BATCH_SIZE = 64
    c.IMAGE_TRAIN = CfgNode({"GPU_IDS": [7], "BATCH_SIZE": 63, "DATASETS": ["ssv2_frames"]})
    s = c.SOLVER
    s.BASE_LR, s.COSINE_END_LR, s.WARMUP_START_LR = 2e-4, 2e-6, 2e-6
    s.MAX_EPOCH, s.OPTIMIZING_METHOD, s.WEIGHT_DECAY = 50, "adamw", 1e-4
    s.BASE_LR_SCALE_NUM_SHARDS = s.COSINE_AFTER_WARMUP = s.ZERO_WD_1D_PARAM = True
    s.CLIP_GRAD_L2NORM = 1.0
    return c
