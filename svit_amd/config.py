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


# Defaults of the hot-path sections, key for key those of slowfast/config/defaults.py:12-1173
# (DATA :473-577, MODEL :290-343, MVIT :345-471, SVIT :1160-1166, SOLVER :580-660, TRAIN :83-140,
# TEST :190-220, IMAGE_TRAIN).  Pinned by tests/golden/cfg.json, which oracle/gen_golden.py dumped
# from the reference's own get_cfg(); tests/test_oracle_golden.py compares every key and type.
_DATA = {
    "DECODING_BACKEND": "pyav", "ENSEMBLE_METHOD": "sum", "INPUT_CHANNEL_NUM": [3, 3],
    "INV_UNIFORM_SAMPLE": False, "MEAN": [0.45, 0.45, 0.45], "MULTI_LABEL": False, "NUM_FRAMES": 8,
    "PATH_LABEL_SEPARATOR": " ", "PATH_PREFIX": "", "PATH_TO_DATA_DIR": "",
    "PATH_TO_PRELOAD_IMDB": "", "RANDOM_FLIP": True, "REVERSE_INPUT_CHANNEL": False,
    "SAMPLING_RATE": 8, "STD": [0.225, 0.225, 0.225], "TARGET_FPS": 30, "TARGET_RES": [28, 28],
    "TEST_CROP_SIZE": 256, "TRAIN_CROP_SIZE": 224, "TRAIN_JITTER_ASPECT_RELATIVE": [],
    "TRAIN_JITTER_MOTION_SHIFT": False, "TRAIN_JITTER_SCALES": [256, 320],
    "TRAIN_JITTER_SCALES_RELATIVE": [], "TRAIN_PCA_EIGVAL": [0.225, 0.224, 0.229],
    "TRAIN_PCA_EIGVEC": [[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.814],
                         [-0.5836, -0.6948, 0.4203]],
    "USE_OFFSET_SAMPLING": False}
_MODEL = {
    "ACT_CHECKPOINT": False, "ARCH": "slowfast", "DROPCONNECT_RATE": 0.0, "DROPOUT_RATE": 0.5,
    "FC_INIT_STD": 0.01, "HEAD_ACT": "softmax", "LOAD_IN_PRETRAIN": "", "LOSS_FUNC": "cross_entropy",
    "MODEL_NAME": "SlowFast", "MULTI_PATHWAY_ARCH": ["slowfast"], "NUM_CLASSES": 400,
    "ROI_HEAD_ACT_DURING_TRAINING": False,
    "SINGLE_PATHWAY_ARCH": ["2d", "c2d", "i3d", "slow", "x3d", "mvit"]}
_MVIT = {
    "ACT_CHECKPOINT": False, "CLS_EMBED_ON": True, "DEPTH": 16, "DIM_MUL": [], "DIM_MUL_IN_ATT": True,
    "DROPOUT_RATE": 0.0, "DROPPATH_RATE": 0.1, "EMBED_DIM": 96, "HEAD_INIT_SCALE": 1.0,
    "HEAD_MUL": [], "IMAGE_KERNEL_FULL_PAD": False, "LAYER_SCALE_INIT_VALUE": 0.0, "MLP_RATIO": 4.0,
    "MODE": "conv", "NORM": "layernorm", "NORM_STEM": False, "NUM_HEADS": 1,
    "OBJECTS_MASKING": False, "PATCH_2D": False, "PATCH_AVG_TEMP": -1, "PATCH_KERNEL": [3, 7, 7],
    "PATCH_PADDING": [2, 4, 4], "PATCH_STRIDE": [2, 4, 4], "POOL_FIRST": False,
    "POOL_KVQ_KERNEL": None, "POOL_KV_IGNORE_111_KERNEL": False, "POOL_KV_STRIDE": None,
    "POOL_KV_STRIDE_ADAPTIVE": None, "POOL_Q_STRIDE": [], "QKV_BIAS": True,
    "REL_POS_SPATIAL": False, "REL_POS_TEMPORAL": False, "REL_POS_ZERO_INIT": False,
    "RESIDUAL_POOLING": True, "SEPARATE_QKV": False, "SEP_POS_EMBED": False, "USE_ABS_POS": True,
    "USE_FIXED_SINCOS_POS": False, "USE_MEAN_POOLING": False, "USE_MLP": False,
    "ZERO_DECAY_POS_CLS": True}
_SVIT = {"LAMBDA_CON": 1.0, "LAMBDA_EDGES": 1.0, "LAMBDA_NODES": 1.0, "O": 4}
_SOLVER = {
    "BASE_LR": 0.1, "BASE_LR_SCALE_NUM_SHARDS": False, "CLIP_GRAD_L2NORM": None,
    "CLIP_GRAD_VAL": None, "COSINE_AFTER_WARMUP": False, "COSINE_END_LR": 0.0, "DAMPENING": 0.0,
    "GAMMA": 0.1, "LRS": [], "LR_POLICY": "cosine", "MAX_EPOCH": 300, "MOMENTUM": 0.9,
    "NESTEROV": True, "OPTIMIZING_METHOD": "sgd", "STEPS": [], "STEP_SIZE": 1, "WARMUP_EPOCHS": 0.0,
    "WARMUP_FACTOR": 0.1, "WARMUP_START_LR": 0.01, "WEIGHT_DECAY": 0.0001, "ZERO_WD_1D_PARAM": False}
_TRAIN = {
    "AUTO_RESUME": True, "BATCH_SIZE": 63, "CHECKPOINT_CLEAR_NAME_PATTERN": [],
    "CHECKPOINT_EPOCH_RESET": False, "CHECKPOINT_FILE_PATH": "", "CHECKPOINT_INFLATE": False,
    "CHECKPOINT_PERIOD": 10, "CHECKPOINT_REPLACE_NAME_PATTERN": [], "CHECKPOINT_TYPE": "pytorch",
    "DATASET": "kinetics", "ENABLE": True, "ENABLE_DOH": False, "EVAL_PERIOD": 10,
    "FORWARD_VIDEO_FRAMES": True, "MIXED_PRECISION": False, "VAL_ONLY": False}
_TEST = {
    "BATCH_SIZE": 8, "CHECKPOINT_FILE_PATH": "", "CHECKPOINT_TYPE": "pytorch", "DATASET": "kinetics",
    "ENABLE": True, "NUM_ENSEMBLE_VIEWS": 10, "NUM_SPATIAL_CROPS": 3, "SAVE_RESULTS_PATH": ""}
_IMAGE_TRAIN = {"BATCH_SIZE": 63, "DATASETS": ["ssv2_frames"], "GPU_IDS": [7]}


def get_cfg():
    """Defaults of every key in the sections the SViT path reads (DATA, MODEL, MVIT, SVIT, SOLVER,
    TRAIN, TEST, IMAGE_TRAIN: complete, as in slowfast/config/defaults.py) plus the handful of
    top-level / BN / DETECTION keys the path touches."""
    c = CfgNode()
    c.NUM_GPUS = 1
    c.NUM_SHARDS = 1
    c.SHARD_ID = 0
    c.RNG_SEED = 1
    c.OUTPUT_DIR = "./tmp"
    c.DIST_BACKEND = "nccl"
    c.DDP_FIND_UNUSED_PARAMETERS = False
    c.LOG_PERIOD = 10
    for name, sec in (("DATA", _DATA), ("MODEL", _MODEL), ("MVIT", _MVIT), ("SVIT", _SVIT),
                      ("SOLVER", _SOLVER), ("TRAIN", _TRAIN), ("TEST", _TEST),
                      ("IMAGE_TRAIN", _IMAGE_TRAIN)):
        c[name] = CfgNode(copy.deepcopy(sec))
    # CONSISTENCY is this build's ONE added key: the switch for the paper's frame-clip consistency
    # loss.  "" keeps the as-released behaviour (the frames pass runs, its output is unused,
    # SURVEY.md sec. 0); "l1" / "l2" weight |obj_desc(video) - obj_desc(frames)| with LAMBDA_CON
    # (losses.py:127-136)
    c.SVIT.CONSISTENCY = ""
    c.DETECTION = CfgNode({"ENABLE": False})
    c.BN = CfgNode({"WEIGHT_DECAY": 0.0})
    return c


def ssv2_cfg(num_frames=16, crop=224, num_gpus=1):
    """get_cfg() + what the reference's configs/ssv2.yaml changes in the hot-path sections (the
    merged tree is pinned by tests/golden/cfg.json; NUM_GPUS there is 8)."""
    c = get_cfg()
    c.NUM_GPUS = num_gpus
    c.RNG_SEED = 0
    c.OUTPUT_DIR = "."
    d = c.DATA
    d.NUM_FRAMES, d.SAMPLING_RATE = num_frames, 2
    d.TRAIN_CROP_SIZE = d.TEST_CROP_SIZE = crop
    d.INPUT_CHANNEL_NUM = [3]
    d.DECODING_BACKEND, d.RANDOM_FLIP, d.USE_OFFSET_SAMPLING = "torchvision", False, True
    d.PATH_TO_DATA_DIR, d.PATH_PREFIX = "/home/datasets/", "/home/datasets/smthsmth/frames"
    d.TRAIN_JITTER_ASPECT_RELATIVE, d.TRAIN_JITTER_SCALES_RELATIVE = [0.75, 1.3333], [0.08, 1.0]
    m = c.MODEL
    m.ARCH, m.MODEL_NAME, m.NUM_CLASSES, m.LOSS_FUNC = "mvit", "SViT", 174, "video_image_loss"
    mv = c.MVIT
    mv.PATCH_PADDING = [1, 3, 3]
    mv.DROPPATH_RATE = 0.4
    mv.DIM_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
    mv.HEAD_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
    mv.POOL_KVQ_KERNEL = [3, 3, 3]
    mv.POOL_KV_STRIDE_ADAPTIVE = [1, 8, 8]
    mv.POOL_Q_STRIDE = [[i, 1, 2, 2] if i in (1, 3, 14) else [i, 1, 1, 1] for i in range(16)]
    mv.REL_POS_SPATIAL = mv.REL_POS_TEMPORAL = True
    mv.USE_ABS_POS = False
    mv.ZERO_DECAY_POS_CLS = False
    c.SVIT.LAMBDA_NODES, c.SVIT.LAMBDA_EDGES, c.SVIT.LAMBDA_CON = 3.7, 0.3, 1.5
    t = c.TRAIN
    t.DATASET, t.MIXED_PRECISION, t.CHECKPOINT_EPOCH_RESET = "ssv2", True, True
    t.CHECKPOINT_PERIOD, t.EVAL_PERIOD = 1, 5
    c.TEST.DATASET, c.TEST.BATCH_SIZE = "ssv2", 64
    s = c.SOLVER
    s.BASE_LR, s.COSINE_END_LR, s.WARMUP_START_LR = 2e-4, 2e-6, 2e-6
    s.MAX_EPOCH, s.OPTIMIZING_METHOD = 50, "adamw"
    s.BASE_LR_SCALE_NUM_SHARDS = s.COSINE_AFTER_WARMUP = s.ZERO_WD_1D_PARAM = True
    s.CLIP_GRAD_L2NORM = 1.0
    return c
