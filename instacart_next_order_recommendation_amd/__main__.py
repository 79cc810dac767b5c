"""`python -m instacart_next_order_recommendation_amd --config configs/inference.yaml` (takes the reference's inference YAML)."""
import sys

from .cli import main

sys.exit(main())
