from .Encoder import Classifier_Module, Deeplabv2, PPMBilinear  # noqa: F401
