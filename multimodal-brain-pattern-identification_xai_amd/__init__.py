"""brainxai -- MI355X-native multimodal brain-pattern training + attribution hot path.

The directory is named after the reference repository (``multimodal-brain-pattern-identification_xai_amd``,
not an importable identifier); ``import brainxai`` (the shim at the repo root) loads it under that name.
"""
from . import _lib, ops                                                       # noqa: F401
from .models import (Attention, Block, EEGNet, EEGNetAttentionDeep, KLDivLoss, MultimodalModel, Spectrogram_Model,   # noqa: F401
                     build_multimodal, set_compute_dtype)
from .explain import (GradCamSweep, expected_gradients, generate_saliency_maps, grad_cam, integrated_gradients, saliency,   # noqa: F401
                      predict_fn, shard_bounds, sharded_sweep)
from .data import (EEGStacker, stack_eeg, EEGMontageStacker, stack_eeg_montage,          # noqa: F401
                   SpectrogramPreprocessor, preprocess_spectrograms, SpectrogramRegionStacker, stack_spectrogram_regions, StagingRing)                                        # noqa: F401
from .train import (FlatAdamW, DataParallel, GraphedTrainStep, AsyncCheckpointer, train_and_validate_combined, train_and_validate_eeg_distributed,   # noqa: F401
                    train_step, train_step_overlapped, overlap_plan, setup, cleanup, create_ddp_model, l2_penalty_, load_checkpoint, load_checkpoint_distributed,
                    save_checkpoint)

__version__ = "0.1.0"
