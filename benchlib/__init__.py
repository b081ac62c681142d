"""The legs of bench.py.  Each leg returns (summary, detail, ok): `summary` = the few scalars that ride on the compact headline line,
`detail` = the leg's full record (written to the detail file), `ok` = its result check."""
