from .reducer import BucketedGradReducer, DataParallelModel, init_dist

__all__ = ['BucketedGradReducer', 'DataParallelModel', 'init_dist']
