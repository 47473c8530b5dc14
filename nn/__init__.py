"""Package alias keeping the reference's import surface (`import nn`, README.md:49-52,
animal2vec_train.py:22): the registered names below resolve to the MI355X-native implementations."""
from animal2vec_amd.config import (  # noqa: F401
    Data2VecMultiConfig, D2vAudioConfig, D2vDecoderConfig, large_config, base_star_config)
from animal2vec_amd.model import Data2VecMultiModel  # noqa: F401
from animal2vec_amd.criterion import ExpandedModelCriterion, ExpandedModelCriterionConfig  # noqa: F401
from animal2vec_amd import checkpoint  # noqa: F401  (fairseq-style .pt wire format)
from animal2vec_amd.finetune import (  # noqa: F401
    Wav2VecCcasFinetune, FinetuneCrossEntropyCriterion, FinetuneHead, FinetuneConfig, FinetuneTrainer, finetune_step,
    sigmoid_focal_loss)
from animal2vec_amd.task import AudioTaskCCAS, AudioConfigCCAS  # noqa: F401
from animal2vec_amd.trainer import PretrainTrainer, OptimConfig, get_annealed_rate  # noqa: F401
from animal2vec_amd.train_routine import animal2vec_audio_main  # noqa: F401
from animal2vec_amd.registry import MODEL_REGISTRY, TASK_REGISTRY, CRITERION_REGISTRY  # noqa: F401
