"""The 17 evaluation regions of public_benchmark/run_benchmark_evaluation.py:110-131 (coordinates are data)."""
REGIONS = {
    'global': ((-90, 90), (0, 360)), 'tropics': ((-20, 20), (0, 360)), 'northern-hemisphere': ((20, 90), (0, 360)),
    'southern-hemisphere': ((-90, -20), (0, 360)), 'europe': ((35, 75), (-12.5, 42.5)),
    'north-america': ((25, 60), (360 - 120, 360 - 75)), 'north-atlantic': ((25, 65), (360 - 70, 360 - 10)),
    'north-pacific': ((25, 60), (145, 360 - 130)), 'east-asia': ((25, 60), (102.5, 150)),
    'ausnz': ((-45, -12.5), (120, 175)), 'arctic': ((60, 90), (0, 360)), 'antarctic': ((-90, -60), (0, 360)),
    'northern-africa': ((5, 32.5), (-12.5, 37.5)), 'southern-africa': ((-30, 5), (12.5, 37.5)),
    'south-america': ((-40, 5), (-75, -45)), 'west-asia': ((15, 60), (42.5, 102.5)),
    'south-east-asia': ((-12.5, 25), (95, 125)),
}
