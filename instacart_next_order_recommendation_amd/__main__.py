"""`python -m instacart_next_order_recommendation_amd` = the reference's `python -m src.inference`."""
from .cli import main

main()
