raise AssertionError('stub reference package core/graph/__init__.py was executed: the MI355X mirror did not shadow it')
