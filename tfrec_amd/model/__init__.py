from .BPR import BPR
from .DNSBPR import DNSBPR
from .APR import APR
from .Pop import Pop
from .ModelFactory import ModelFactory
