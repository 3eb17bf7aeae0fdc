from .test import collect_results, encode_poly_results, multi_gpu_test, single_gpu_test
from .train import train_detector

__all__ = ['train_detector', 'single_gpu_test', 'multi_gpu_test', 'collect_results', 'encode_poly_results']
