import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_built():
    import orc

    if not (orc.ORACLE_DIR / "liboracle.so").exists() or not (orc.ORACLE_DIR / "liboracle_libm.so").exists():
        orc.build_oracle()
    return True
