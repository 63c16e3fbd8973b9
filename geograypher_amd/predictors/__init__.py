from geograypher_amd.predictors.segmentor import Segmentor
from geograypher_amd.predictors.derived_segmentors import ArrayLabelSegmentor, LookUpSegmentor

__all__ = ["Segmentor", "LookUpSegmentor", "ArrayLabelSegmentor"]
