raise AssertionError('stub reference package core/utils/__init__.py was executed: the MI355X mirror did not shadow it')
