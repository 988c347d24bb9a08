from .evaluator import FoldOutEvaluator
