from .dataiterator import DataIterator
from .dataset import Dataset
from .device_sampler import DeviceEpochSampler
from .splitter import Splitter
