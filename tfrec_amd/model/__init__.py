from .BPR import BPR
from .DNSBPR import DNSBPR
from .APR import APR
from .ModelFactory import ModelFactory
