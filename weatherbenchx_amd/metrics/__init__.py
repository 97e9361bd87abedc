"""Metric / Statistic plugin surface (mirror of weatherbenchX/metrics)."""
