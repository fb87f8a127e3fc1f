"""One object for the reference's hyper-parameters.

The reference scatters them: ``main`` hard-codes the run (train.py:358-373: num_epochs 8, batch_size 1024,
learning_rate 1.0, margin 0.8, LARSOptimizer, check_stop_epoch 3, best_eval_dist 1.0, eval_per_epoch 100,
require_improve_num 40), ``Trainer._build_model`` the graph (train.py:212-222: output_size 256,
learning_rate_decay_examples 1 000 000, learning_rate_decay 0.96, clip_gradient_norm 0, regularization_penalty 0),
the model name comes from a flag (train.py:38-44, 351) and the hidden width is inside models.py:59.  ``TrainConfig``
carries exactly these names with exactly these defaults (+ the build's own switches), round-trips through JSON, and
builds the ``TrainStep`` / ``Trainer`` pair ("Config / flags")."""
import dataclasses
import json


@dataclasses.dataclass
class TrainConfig:
    # train.py:358-373 (main)
    num_epochs: int = 8
    batch_size: int = 1024
    learning_rate: float = 1.0
    margin: float = 0.8
    optimizer: str = "lars"                   # tf.contrib.opt.LARSOptimizer (train.py:354); "adam" = build_graph's default (train.py:82)
    check_stop_epoch: float = 3
    best_eval_dist: float = 1.0
    eval_per_epoch: int = 100
    require_improve_num: int = 40
    # train.py:212-222 (_build_model)
    output_size: int = 256
    learning_rate_decay_examples: int = 1000000
    learning_rate_decay: float = 0.96
    clip_gradient_norm: float = 0.0
    regularization_penalty: float = 0.0
    # train.py:38-44 (flags), models.py:59
    model: str = "VNet"
    hidden_size: int = 5000
    train_dir: str = ""
    checkpoint_dir: str = ""
    # the build's own switches (no reference counterpart)
    mode: str = "uniform"                     # the reference's negative rule (inputs.py:125-127); "inbatch" / "semihard": BASELINE configs 1 / 2
    precision: str = "f32x3"                  # "f32": the fp32 MFMA; "bf16": BASELINE config 4 (fp16 catalogue)
    seed: int = 1234
    weight_seed: int = 42

    def to_json(self, path=None):
        s = json.dumps(dataclasses.asdict(self), indent=1)
        if path:
            with open(path, "w") as f:
                f.write(s)
        return s

    @classmethod
    def from_json(cls, text_or_path):
        try:
            d = json.loads(text_or_path)
        except ValueError:
            with open(text_or_path) as f:
                d = json.load(f)
        unknown = set(d) - {f.name for f in dataclasses.fields(cls)}
        if unknown:
            raise ValueError("unknown configuration keys: %s" % sorted(unknown))
        return cls(**d)

    def train_step(self, table, pairs, device="cuda:0", **overrides):
        """The ``TrainStep`` of this configuration (``overrides``: exchange / grad_sync hooks, use_graph, ...)."""
        from . import train
        if self.model != "VNet":
            raise ValueError("TrainStep runs the VNet tower; the fusion towers are fusion.FusionTrainStep(%r, ...)" % self.model)
        kw = dict(output_size=self.output_size, hidden_size=self.hidden_size, margin=self.margin, mode=self.mode,
                  optimizer=self.optimizer, base_learning_rate=self.learning_rate,
                  learning_rate_decay_examples=self.learning_rate_decay_examples,
                  learning_rate_decay=self.learning_rate_decay, seed=self.seed, weight_seed=self.weight_seed, device=device,
                  precision=self.precision, clip_gradient_norm=self.clip_gradient_norm,
                  regularization_penalty=self.regularization_penalty)
        kw.update(overrides)
        return train.TrainStep(table, pairs, self.batch_size, **kw)

    def trainer(self, train_step, n_pairs, eval_features=None, eval_cowatches=None, **overrides):
        """The reference's loop and model-selection policy (train.py:177-336) around ``train_step``."""
        from . import train
        kw = dict(checkpoint_dir=self.checkpoint_dir or None, eval_features=eval_features, eval_cowatches=eval_cowatches,
                  check_stop_epoch=self.check_stop_epoch, best_eval_dist=self.best_eval_dist,
                  eval_per_epoch=self.eval_per_epoch, require_improve_num=self.require_improve_num)
        kw.update(overrides)
        return train.Trainer(train_step, self.num_epochs, n_pairs, **kw)
