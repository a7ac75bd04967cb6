"""bpl -- MI355X-native drop-in for anguswilliams91/bpl-next's Dixon-Coles predictors.

Same public surface as the reference's bpl/__init__.py:4-7 for the models on the hot
path; the numpyro/JAX machinery underneath is replaced by libbplhip.so (HIP, gfx950).
"""
__version__ = "0.2.0"

from bpl.dixon_coles import DixonColesMatchPredictor
from bpl.extended_dixon_coles import ExtendedDixonColesMatchPredictor
from bpl.neutral_dixon_coles import NeutralDixonColesMatchPredictor
from bpl.neutral_dixon_coles_WC import NeutralDixonColesMatchPredictorWC

__all__ = ["DixonColesMatchPredictor", "ExtendedDixonColesMatchPredictor",
           "NeutralDixonColesMatchPredictor", "NeutralDixonColesMatchPredictorWC"]
