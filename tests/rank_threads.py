"""Test-side name of chase_amd/rank_threads.py (the ranks of a grid as threads of one process): the module moved into the
package in round 4 because `bench.py --ranks threads` uses it too."""
from chase_amd.rank_threads import *  # noqa: F401,F403
from chase_amd.rank_threads import _World, ROW, COL, TIMEOUT  # noqa: F401
