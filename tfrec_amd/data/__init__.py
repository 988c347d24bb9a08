from .dataiterator import DataIterator
from .dataset import Dataset
