from .train import train_detector

__all__ = ['train_detector']
