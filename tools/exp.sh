python tools/row_stats.py 22
