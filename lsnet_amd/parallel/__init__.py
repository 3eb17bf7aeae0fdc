from .data_container import DataContainer, collate, scatter
from .reducer import BucketedGradReducer, DataParallelModel, init_dist

__all__ = ['BucketedGradReducer', 'DataParallelModel', 'init_dist', 'DataContainer', 'collate', 'scatter']
