"""`from resources import generator` as main.py:2 does: only the CSV writer main.py uses
(generator.write_csv, main.py:109; reference resources/generator.py:60-77) is provided here --
the synthetic data generator itself is outside this build's scope."""
import csv
import os


def write_csv(filename, header, data):
    """Write ./output/<filename>.csv relative to the working directory: an optional header row
    (None = no header), then one row per entry of `data`."""
    os.makedirs("output", exist_ok=True)
    with open(os.path.join("output", filename + ".csv"), "w", newline="") as fh:
        out = csv.writer(fh)
        if header is not None:
            out.writerow(header)
        out.writerows(data)
