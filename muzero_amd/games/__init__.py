"""Game definitions with the reference's module layout: games/<name>.make_config() (utils.py:128-134)."""
