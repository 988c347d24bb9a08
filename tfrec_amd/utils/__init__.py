from .tools import timer, typeassert, csr_to_user_dict, csr_to_user_item_pair, random_choice, argmax_top_k
from .logger import Logger
from .params_io import save_params, load_params, save_checkpoint, load_checkpoint
