"""ncahip -- MI355X-native NCA update loop behind the reference's Python operator surface.

Drop-in modules (same class names / signatures / state_dict keys as the reference):
    ncahip.nca                  ConditionedNCA, UpdateNet      (EncoderConditioning/nca.py)
    ncahip.encoder              ImageEncoder                   (EncoderConditioning/encoder.py)
    ncahip.sample_pool          SamplePool                     (EncoderConditioning/sample_pool.py)
    ncahip.trainer              NCATrainer                     (EncoderConditioning/trainer.py)
    ncahip.conditioned_trainer  ConditionedNCATrainer          (EncoderConditioning/conditioned_trainer.py)
    ncahip.models.dynca         DyNCA, EdgeExtractor, CPE2D    (ConditioneDyNCA/models/dynca.py)
    ncahip.models.dynca_extra   DyNCA (state-concat variant)   (ExtraChannels/models/dynca.py)

The hot path (perception stencil + per-pixel MLP + stochastic mask + residual [+ alive mask +
clamp]) runs in hand-written HIP kernels from libncahip.so through a plain C ABI
(include/ncahip.h); there is no CPU or eager-PyTorch fallback for it.
"""
__version__ = "0.1.0"
